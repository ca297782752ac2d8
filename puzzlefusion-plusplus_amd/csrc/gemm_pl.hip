// split-f16 x3 GEMM on pre-split planes, LDS-DMA staged, software-pipelined fragment reads (arithmetic: gemm.hip).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  A and W given as fp16 hi/lo planes with the contraction contiguous
//   (K % 32 == 0; rows zero-padded by the producer).  Three v_mfma_f32_32x32x16_f16 per 16-deep step and
//   accumulator, in the order lo.hi, hi.lo, hi.hi — the order of gemm_f16x3_kernel, so the two kernels agree
//   bit for bit.
//
// Why another loop.  gemm_f16x3_kernel stages through registers (load -> split -> ds_write -> barrier -> ds_read)
// and restarts its LDS -> register pipeline at every 16-deep step; it runs the matrix pipe at 30-36 %
// (DESIGN.md §3.1).  Here
//   * every global load is an LDS-DMA (global_load_lds_dwordx4: no staging registers, no conversion, no ds_write);
//     NS stages of BK = 32 (64-byte plane rows), one raw s_barrier per K-tile, counted vmcnt — loads stay in
//     flight across barriers (cdna guide §5, T3/T4);
//   * the 16-byte chunks of a row are permuted on the SOURCE side (chunk ^= (row >> 2) & 3) and read back with
//     the same XOR, so the lane-linear DMA image is bank-conflict free for ds_read_b128 (guide rule 21);
//   * a wave owns (32 MT) x (32 NT) of the tile (128x64 in the 256x256 tile: 12 fragment reads per 24 MFMAs) and
//     keeps its fragments double-buffered in registers: the reads of half-step h+1 — across K-tile boundaries too —
//     are issued before the MFMAs of half-step h, so the matrix instructions of a wave never wait for LDS;
//   * two waves per SIMD (8-wave workgroups, or two 4-wave workgroups per CU) cover each other's barrier waits.
// Operand layouts (template AK / WK).  Row-major: the operand's rows are output rows / columns and the contraction runs
// along a row (forward GEMMs; 64-byte plane rows per K-tile, 16-byte chunks XOR-swizzled).  K-major: the operand is stored
// [contraction][rows] — W in dX = dY . W, and both dY and X in dW = dY^T . X — and is NOT copied transposed: its K-tile is
// staged as 32 contraction rows x BM (BN) columns, and fragments are fetched with ds_read_b64_tr_b16, which hands each lane
// of a 16-lane group the 4 contraction-consecutive halfs of its own column out of a [4][16] block (lane -> element map
// probed on gfx950: tools/gemm_lab/trprobe.hip).  Two such reads make one MFMA operand.  The 16-byte chunks of a
// contraction row are rotated by 4 * (row & 3) (64-column tiles: 4 * ((row >> 1) & 1)) on the DMA source side so that the
// four rows a 32-lane group touches fall into disjoint banks.
// Split-K (grid.y > 1) and accumulate mode add alpha * acc into C with fp32 atomics (weight / input gradients).
// Fragment reads are inline asm (hipcc drains vmcnt(0) in front of every ds_read it can see while an LDS-DMA is
// in flight); the waits name every destination register, which is what orders the MFMAs behind them.
// One __shared__ object only (a second one makes hipcc drain the DMA queue at every step, guide §5).
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "gemm_common.h"

namespace pfpp_gemm_detail {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace pl {

// source of the DMA lanes whose contraction row lies beyond the operand's last valid row (k-major operands with a ragged
// contraction length, e.g. 3,850 tokens): they read zeros from here instead of whatever follows the tensor
__device__ __attribute__((aligned(64))) char pl_zero_row[64] = {0};

constexpr int BK = 32;
thread_local int64_t p_ws_bytes = 0;     // capacity of the caller's K-split workspace for the launch being dispatched
thread_local char last_kernel[96] = "";  // template instantiation of the most recent plane-GEMM launch of this thread (pfpp_last_gemm_kernel)

__device__ __forceinline__ void glds16(const char* gsrc, uint32_t ldst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(uintptr_t)ldst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ half8 lds_rd(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  half8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// MT x NT 32x32 tiles per wave, WM x WN waves.  HS = row tiles per half-step (the fragment double buffer holds HS row tiles
// and NT column tiles): HS == MT for one half-step per 16-deep step, MT / 2 for two.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ f32x4 lds_rd_f4(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// transposing read: within a 16-lane group, lane j receives element j of each of the four 16-half rows the group's lanes
// address (lanes 4r .. 4r+3 supply row r, 4 halfs each)
template <int OFF>
__device__ __forceinline__ half4 lds_rd_tr(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  half4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int MT, int NT, int WM, int WN, int NS, bool X1 = false>
struct Cfg {
  static constexpr int NPL = X1 ? 1 : 2;                  // planes per operand in LDS: single-pass fp16 stages the hi planes only
  static constexpr int HS = (MT == 4 && NT == 2) ? 2 : MT;
  static constexpr int WPS = (MT * NT >= 16) ? 1 : 2;     // waves per SIMD the register budget allows (512 / 256 registers)
  static constexpr int NW = WM * WN;
  static constexpr int NTHR = 64 * NW;
  static constexpr int BM = 32 * MT * WM;
  static constexpr int BN = 32 * NT * WN;
  static constexpr int PLANE_A = BM * 64;                 // bytes: BM rows of 32 halfs
  static constexpr int PLANE_W = BN * 64;
  static constexpr int STAGE = NPL * (PLANE_A + PLANE_W);
  static constexpr int NP = STAGE / 1024;                 // 1 KiB DMA pieces (16 rows of one plane) per stage
  static constexpr int NPW = NP / NW;                     // pieces per wave
  static constexpr size_t SMEM = (size_t)NS * STAGE;
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  static_assert((MT == 1 || MT == 2 || MT == 4) && (NT == 1 || NT == 2 || NT == 4), "wave tile: 1, 2 or 4 tiles each way");
};

// Epilogue with 16-byte stores.  An accumulator tile holds, per lane, runs of 4 consecutive ROWS of one column, so storing it
// directly takes one 4-byte store per element: 256 bytes per wave instruction, and the store issue (not the bandwidth) bounds
// the epilogue (measured 3850x1536x512: 9 of 30 us; cdna guide T21).  Here every wave passes its tiles through a private
// 8 KB LDS patch (the DMA ring is dead by then): ds_write_b32 in accumulator order, ds_read_b128 along the rows, and
// each store instruction then writes 4 rows x 256 contiguous bytes.  Covers bias / BN scale+shift / alpha / activation /
// residual / GEGLU and fp32 or split-plane output; pooling and BatchNorm statistics keep the generic epilogue.
template <int MT, int NT>
__device__ __forceinline__ void epilogue_wide(const GemmP& p, f32x16 (&acc)[MT][NT], int row_w, int col_w, int lane,
                                              int64_t c_off, int64_t v_off, uint32_t patch) {
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* bias = p.bias ? p.bias + v_off : nullptr;
  const float* scale = p.scale ? p.scale + v_off : nullptr;
  const float* shift = p.shift ? p.shift + v_off : nullptr;
  const float alpha = p.alpha;
  const bool geglu = p.act == PFPP_ACT_GEGLU;
  constexpr int ROWB = NT * 128;                 // bytes per patch row: NT x 32 floats
  float sc[NT], sh[NT], st_s[NT], st_q[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col_w + j * 32 + l31;
    sc[j] = 1.0f; sh[j] = 0.0f; st_s[j] = 0.0f; st_q[j] = 0.0f;
    if (col < p.N) {
      if (scale) { sc[j] = scale[col]; sh[j] = shift[col]; }
      else if (bias) sh[j] = bias[col];
    }
  }
  // read side: lane t takes 4 floats at column 4 * (t % (8 NT)) of rows t / (8 NT) + k * (64 / (8 NT))
  constexpr int LPR = 8 * NT;                    // lanes per row
  constexpr int RPI = 64 / LPR;                  // rows per read instruction
  const int rcol = (lane % LPR) * 4, rrow = lane / LPR;
  const int out_cols = geglu ? NT * 16 : NT * 32;
  const int ocol0 = geglu ? (col_w >> 1) : col_w;
  const int n_out = geglu ? (p.N >> 1) : p.N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x16 t = acc[i][j];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = t[e] * alpha;
        t[e] = scale ? __builtin_fmaf(v, sc[j], sh[j]) : v + sh[j];
      }
      if (p.stats) {      // BatchNorm batch statistics of the pre-activation (same sums as the generic epilogue)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          const float v = row < p.M ? t[e] : 0.0f;
          st_s[j] += v;
          st_q[j] += v * v;
        }
      }
      int pc = j * 32 + l31;                     // patch column
      if (NT >= 2 && geglu) {
        if ((j & 1) == 0) { acc[i][j] = t; continue; }
        t = act_tile(t, PFPP_ACT_GELU);
        const f32x16 u = acc[i][j > 0 ? j - 1 : 0];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = u[e] * t[e];
        pc = (j >> 1) * 32 + l31;
      } else {
        t = act_tile(t, p.act);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        asm volatile("ds_write_b32 %0, %1" ::"v"(patch + r * ROWB + pc * 4), "v"(t[e]) : "memory");
      }
    }
    // all row reads of the tile in flight, one wait (the fragment buffers are dead here: registers are free)
    f32x4 vv[32 / RPI];
#pragma unroll
    for (int k = 0; k < 32 / RPI; ++k)
      asm volatile("ds_read_b128 %0, %1" : "=v"(vv[k]) : "v"(patch + (rrow + k * RPI) * ROWB + rcol * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 32 / RPI; k += 4)        // the wait orders the uses: name every destination behind it
      asm volatile("" : "+v"(vv[k]), "+v"(vv[k + 1]), "+v"(vv[k + 2]), "+v"(vv[k + 3]));
#pragma unroll
    for (int k = 0; k < 32 / RPI; ++k) {
      const int r = rrow + k * RPI;
      float4 v = make_float4(vv[k][0], vv[k][1], vv[k][2], vv[k][3]);
      const int row = row_w + i * 32 + r;
      const int col = ocol0 + rcol;
      if (row < p.M && rcol < out_cols && col < n_out) {
        const int64_t idx = c_off + (int64_t)row * p.ldc + col;
        if (p.residual) {
          const float4 q = *reinterpret_cast<const float4*>(p.residual + c_off + (int64_t)row * p.ldr + col);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (p.Chi) {
          typedef _Float16 half4v __attribute__((ext_vector_type(4)));
          half4v hi, lo;
          PFPP_SPLIT_TO(v.x, hi[0], lo[0]); PFPP_SPLIT_TO(v.y, hi[1], lo[1]);
          PFPP_SPLIT_TO(v.z, hi[2], lo[2]); PFPP_SPLIT_TO(v.w, hi[3], lo[3]);
          *reinterpret_cast<half4v*>(reinterpret_cast<_Float16*>(p.Chi) + idx) = hi;
          *reinterpret_cast<half4v*>(reinterpret_cast<_Float16*>(p.Clo) + idx) = lo;
        } else {
          *reinterpret_cast<float4*>(p.C + idx) = v;
        }
      }
    }
  }
  if (p.stats) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col_w + j * 32 + l31;
      const float a = st_s[j] + __shfl_xor(st_s[j], 32), b = st_q[j] + __shfl_xor(st_q[j], 32);
      if (lhi == 0 && col < p.N) {
        double* st = p.stats + (size_t)(blockIdx.x % p.stats_copies) * 2 * p.N;
        unsafeAtomicAdd(st + col, (double)a);
        unsafeAtomicAdd(st + p.N + col, (double)b);
      }
    }
  }
}

// One workgroup = one output tile.  NS-stage DMA ring; see the header for the schedule.
// EXT: the (tile, K chunk) of this workgroup comes from the caller (ext_tile of the problem's row-major tile list, whole contraction)
// instead of blockIdx — the grouped weight-gradient launch, where a workgroup first finds its problem in a table.
template <int MT, int NT, int WM, int WN, int NS, bool AK, bool WK, int DBG, bool AF = false, bool X1 = false, bool CS = false, bool EXT = false>
__device__ __forceinline__ void pl_body(const GemmP& p, int ext_tile = 0) {
  static_assert(!(AF && AK), "an fp32 A operand is row-major");
  static_assert(!CS || (AK && !X1), "column sums ride with a k-major split A operand (dW = dY^T . X)");
  static_assert(!(AF && X1), "the single-pass mode takes pre-split operands");
  using C = Cfg<MT, NT, WM, WN, NS, X1>;
  constexpr int NPL = C::NPL;
  constexpr int BM = C::BM, BN = C::BN, NPW = C::NPW, STAGE = C::STAGE;
  constexpr int HS = C::HS;
  constexpr int NH = MT / HS;         // half-steps per 16-deep step
  constexpr int NMMA = (X1 ? 1 : 3) * HS * NT;   // MFMAs per half-step (single pass: hi.hi only)
  constexpr int RA = (AK ? 2 : 1) * NPL * HS, RB = (WK ? 2 : 1) * NPL * NT;   // fragment reads of a half-step's A tiles / a step's W tiles
  constexpr int CPR_A = BM / 8, CPR_W = BN / 8;                  // 16-byte chunks per contraction row of a k-major tile
  extern __shared__ __align__(1024) char pl_smem[];
  if constexpr (!AF && !(AK && WK)) pfpp_chain_prio();      // forward / input-gradient forms: the step's dependency chain

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  // 1-D grid of (K chunk, tile) pairs, chunk-major after the XCD remap: an XCD walks one K range over many tiles
  const int wg = EXT ? ext_tile : remap_tile(blockIdx.x, gridDim.x);
  const int n_tiles = p.tiles_m * p.tiles_n;
  const int split = EXT ? 0 : wg / n_tiles;
  const int tile = wg - split * n_tiles;
  int tm, tn;
  tile_coords(p, tile, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_all = p.K / BK;
  const int nk_base = nk_all / p.split_k, nk_rem = nk_all - nk_base * p.split_k;
  const int kt0 = split * nk_base + min(split, nk_rem);      // first K-tile of this workgroup
  const int nk = nk_base + (split < nk_rem ? 1 : 0);
  const int z = EXT ? 0 : blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const int64_t a_offz = z0 * p.sA0 + z1 * p.sA1;
  const int64_t w_offz = z0 * p.sW0 + z1 * p.sW1;
  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)pl_smem;

  // ---- DMA sources: piece q = wave + NW*j of a stage = LDS bytes [q KiB, (q+1) KiB) of one plane ----------------------------
  // row-major operand: lane i of a piece lands at row (i >> 2) of the piece's 16 rows, physical chunk i & 3, which holds the
  //   row's logical 16-byte chunk (i & 3) ^ ((row >> 2) & 3); the next K-tile is 64 bytes further along the row
  // k-major operand: plane chunk ci = contraction row ci / CPR, slot ci % CPR, which holds column chunk
  //   (slot - 4 * rot(row)) mod CPR of that row; the next K-tile is 32 rows further down
  const char* src[NPW];
  uint32_t dst[NPW];
  const int64_t adv_a = AK ? (int64_t)32 * p.lda * 2 : (AF ? 128 : 64), adv_w = WK ? (int64_t)32 * p.ldw * 2 : 64;
  bool piece_a[NPW];
  int piece_krow[NPW];                 // k-major pieces: this lane's contraction row within the K-tile
  const int k_tail = p.k_valid - (nk_all - 1) * BK;      // valid rows of the last K-tile (BK when the contraction is not ragged)
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int q = wave + C::NW * j;
    const int o = q * 1024;
    const bool is_a = o < (AF ? 2 : NPL) * C::PLANE_A;
    piece_a[j] = is_a;
    const int o2 = is_a ? o : o - (AF ? 2 : NPL) * C::PLANE_A;
    const int psz = is_a ? C::PLANE_A : C::PLANE_W;
    const bool lo = NPL == 2 && o2 >= psz;
    const int ci = ((lo ? o2 - psz : o2) >> 4) + lane;           // 16-byte chunk index within the plane
    const _Float16* base = reinterpret_cast<const _Float16*>(is_a ? (lo ? p.Alo : p.Ahi) : (lo ? p.Wlo : p.Whi));
    int64_t eoff;
    piece_krow[j] = 0;
    auto kmajor_off = [&](int cpr, int g0, int lim, int64_t ld) {
      const int krow = ci / cpr, slot = ci % cpr;
      piece_krow[j] = krow;
      const int rot = cpr >= 16 ? (krow & 3) : ((krow >> 1) & 1);
      const int nc = (slot - 4 * rot + cpr) % cpr;
      const int col = min(g0 + 8 * nc, lim - 8);
      return (int64_t)krow * ld + col;
    };
    auto rowmajor_off = [&](int g0, int lim, int64_t ld) {
      const int row = ci >> 2;
      const int chunk = (ci & 3) ^ ((row >> 2) & 3);
      return (int64_t)min(g0 + row, lim - 1) * ld + chunk * 8;
    };
    if (AF && is_a) {
      // fp32 rows of 128 bytes (32 k): plane-region chunk index cf = row * 8 + physical chunk, logical = physical ^ ((row >> 1) & 7)
      const int cf = (o >> 4) + lane;
      const int row = cf >> 3;
      const int chunk = (cf & 7) ^ ((row >> 1) & 7);
      src[j] = reinterpret_cast<const char*>(p.A + a_offz + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 4) + (int64_t)kt0 * adv_a;
      dst[j] = lds0 + o;
      continue;
    }
    if (is_a) eoff = a_offz + (AK ? kmajor_off(CPR_A, m0, p.M, p.lda) : rowmajor_off(m0, p.M, p.lda));
    else eoff = w_offz + (WK ? kmajor_off(CPR_W, n0, p.N, p.ldw) : rowmajor_off(n0, p.N, p.ldw));
    src[j] = reinterpret_cast<const char*>(base + eoff) + (int64_t)kt0 * (is_a ? adv_a : adv_w);
    dst[j] = lds0 + o;
  }
  auto piece_src = [&](int kt, int j) {
    const char* s_ = src[j] + (int64_t)kt * (piece_a[j] ? adv_a : adv_w);
    if constexpr (AK || WK) {
      if (k_tail < BK && kt0 + kt == nk_all - 1 && (piece_a[j] ? AK : WK) && piece_krow[j] >= k_tail) s_ = pl_zero_row;
    }
    return s_;
  };
  auto issue = [&](int kt, uint32_t st_off) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) glds16(piece_src(kt, j), dst[j] + st_off);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  // CS: the workgroups of tile column 0 also sum their A operand over the contraction (sum_k dY[k][m] = the bias gradient):
  // v_dot2c_f32_f16 with ones on the fragments the MFMAs consume anyway, issued in the shadow of the first MFMAs of a half-step
  const bool do_cs = CS && tn == 0 && wn == 0 && (!EXT || p.csum != nullptr);
  float bsum[CS ? MT : 1];
#pragma unroll
  for (int i = 0; i < (CS ? MT : 1); ++i) bsum[i] = 0.0f;
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  const _Float16 cs_w = do_cs ? (_Float16)1.0f : (_Float16)0.0f;
  const half2v cs_one = {cs_w, cs_w};

  // ---- fragment addressing -----------------------------------------------------------------------------------------------
  // row-major: lane (l31, lhi) reads row l31 of a 32-row tile, logical chunk 2*s + lhi; one address per 16-deep step s
  // k-major: lane (q = lane >> 4, j = lane & 15) reads 8 bytes at contraction row 16 s + 8 (q >> 1) + 4 t + (j >> 2), column
  //   32 tile + 16 (q & 1) + 4 (j & 3); one address per tile of the wave, (s, t) go into the offset field
  const int sw = (l31 >> 2) & 3;
  constexpr int NAD_A = AK ? MT : (AF ? 4 : 2), NAD_W = WK ? NT : 2;
  uint32_t a_ad[NAD_A], w_ad[NAD_W];
  auto kmajor_ad = [&](int cpr, int it) {
    const int q = lane >> 4, j = lane & 15;
    const int krow = 8 * (q >> 1) + (j >> 2);
    const int rot = cpr >= 16 ? ((j >> 2) & 3) : (((j >> 2) >> 1) & 1);
    const int nc = 4 * it + 2 * (q & 1) + ((j & 3) >> 1);
    return (uint32_t)((krow * cpr + ((nc + 4 * rot) % cpr)) * 16 + (j & 1) * 8);
  };
  if constexpr (AK) {
#pragma unroll
    for (int ii = 0; ii < MT; ++ii) a_ad[ii] = lds0 + kmajor_ad(CPR_A, wm * MT + ii);
  } else if constexpr (AF) {
    // fp32 tile: row l31, 16-byte chunks 4 s + 2 lhi and + 1 (8 k-values), physical = logical ^ ((row >> 1) & 7)
    const int swf = (l31 >> 1) & 7;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      a_ad[2 * s2] = lds0 + (wm * 32 * MT + l31) * 128 + (((4 * s2 + 2 * lhi) ^ swf) << 4);
      a_ad[2 * s2 + 1] = lds0 + (wm * 32 * MT + l31) * 128 + (((4 * s2 + 2 * lhi + 1) ^ swf) << 4);
    }
  } else {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) a_ad[s2] = lds0 + (wm * 32 * MT + l31) * 64 + (((2 * s2 + lhi) ^ sw) << 4);
  }
  // fp32 A with the train-mode BatchNorm + ReLU of the previous layer fused in: (a_mul, a_add) [K] staged in LDS behind the ring
  const uint32_t aff0 = lds0 + NS * STAGE;
  if constexpr (AF) {
    float* affp = reinterpret_cast<float*>(pl_smem + NS * STAGE);
    for (int k = tid; k < 256; k += C::NTHR) {
      affp[k] = (p.a_mul && k < p.K) ? p.a_mul[k] : 1.0f;
      affp[256 + k] = (p.a_add && k < p.K) ? p.a_add[k] : 0.0f;
    }
    __syncthreads();       // before any LDS-DMA is in flight: a plain barrier is enough here
  }
  if constexpr (WK) {
#pragma unroll
    for (int jj = 0; jj < NT; ++jj) w_ad[jj] = lds0 + (AF ? 2 : NPL) * C::PLANE_A + kmajor_ad(CPR_W, wn * NT + jj);
  } else {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) w_ad[s2] = lds0 + (AF ? 2 : NPL) * C::PLANE_A + (wn * 32 * NT + l31) * 64 + (((2 * s2 + lhi) ^ sw) << 4);
  }
  // a k-major fragment arrives as two 8-byte halves (contraction rows +0..3 and +4..7)
  struct FA { half8 h[HS], l[HS]; half4 h2[AK ? HS : 1][2], l2[AK ? HS : 1][2]; f32x4 r[AF ? HS : 1][2]; f32x4 am[2], aa[2]; };
  struct FB { half8 h[NT], l[NT]; half4 h2[WK ? NT : 1][2], l2[WK ? NT : 1][2]; };
  FA fa[2];
  FB fb[2];
  // DBG (lab builds, -DPFPP_PL_LAB): 1 no epilogue, 2 no DMA after the prologue, 4 no barrier / DMA wait, 8 no fragment reads,
  // 16 no fragment waits
  constexpr bool dbg_noread = DBG & 8, dbg_nowait = DBG & 16;
  // Fragment reads one at a time, each in its own gap between two MFMAs.  Row-major: q < HS: hi plane of row tile q, then
  // the lo planes.  K-major: q = 4 * tile + 2 * plane + t.
  int kt_aff = 0;          // AF: K-tile the next fragment reads belong to (selects the slice of the affine table)
  auto rd_a = [&](FA& f, uint32_t st, auto s_c, auto mh_c, auto q_c) {
    constexpr int s = decltype(s_c)::value, mh = decltype(mh_c)::value, q = decltype(q_c)::value;
    if constexpr (dbg_noread) return;
    if constexpr (AK) {
      constexpr int t = q / (2 * NPL), pl_ = (q >> 1) % NPL, hf = q & 1;
      constexpr int off = pl_ * C::PLANE_A + (16 * s + 4 * hf) * CPR_A * 16;
      const uint32_t ad = a_ad[HS * mh + t] + st;
      if constexpr (pl_ == 0) f.h2[t][hf] = lds_rd_tr<off>(ad); else f.l2[t][hf] = lds_rd_tr<off>(ad);
    } else if constexpr (AF) {
      // q < HS: first 16 bytes of row tile q's 8 k-values, then the second 16 bytes; the affine pair of this 16-deep step
      // rides with the first two reads
      constexpr int t = q % HS, hf = q / HS;
      f.r[t][hf] = lds_rd_f4<(HS * mh + t) * 4096>(a_ad[2 * s + hf] + st);
      if constexpr (t == 0) {
        f.am[hf] = lds_rd_f4<0>(aff0 + kt_aff * 128 + (16 * s + 8 * lhi + 4 * hf) * 4);
        f.aa[hf] = lds_rd_f4<1024>(aff0 + kt_aff * 128 + (16 * s + 8 * lhi + 4 * hf) * 4);
      }
    } else {
      const uint32_t ad = a_ad[s] + st;
      constexpr int t = q % HS;
      if constexpr (q < HS) f.h[t] = lds_rd<(HS * mh + t) * 2048>(ad);
      else f.l[t] = lds_rd<C::PLANE_A + (HS * mh + t) * 2048>(ad);
    }
  };
  auto rd_b = [&](FB& f, uint32_t st, auto s_c, auto q_c) {
    constexpr int s = decltype(s_c)::value, q = decltype(q_c)::value;
    if constexpr (dbg_noread) return;
    if constexpr (WK) {
      constexpr int t = q / (2 * NPL), pl_ = (q >> 1) % NPL, hf = q & 1;
      constexpr int off = pl_ * C::PLANE_W + (16 * s + 4 * hf) * CPR_W * 16;
      const uint32_t ad = w_ad[t] + st;
      if constexpr (pl_ == 0) f.h2[t][hf] = lds_rd_tr<off>(ad); else f.l2[t][hf] = lds_rd_tr<off>(ad);
    } else {
      const uint32_t ad = w_ad[s] + st;
      constexpr int t = q % NT;
      if constexpr (q < NT) f.h[t] = lds_rd<t * 2048>(ad);
      else f.l[t] = lds_rd<C::PLANE_W + t * 2048>(ad);
    }
  };
  // all outstanding LDS reads have landed; names the fragments the following MFMAs consume (k-major: and joins their halves)
  auto join = [](const half4 x, const half4 y) { return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7); };
  auto wait_a = [&](FA& a) {
    if constexpr (dbg_nowait) return;
    if constexpr (X1 && AK) {
      static_for<HS>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h2[t][0]), "+v"(a.h2[t][1]));
        a.h[t] = join(a.h2[t][0], a.h2[t][1]);
      });
    } else if constexpr (X1) {
      if constexpr (HS == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]));
      else if constexpr (HS == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.h[1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.h[1]), "+v"(a.h[2]), "+v"(a.h[3]));
    } else if constexpr (AK) {
      static_for<HS>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h2[t][0]), "+v"(a.h2[t][1]), "+v"(a.l2[t][0]), "+v"(a.l2[t][1]));
        a.h[t] = join(a.h2[t][0], a.h2[t][1]);
        a.l[t] = join(a.l2[t][0], a.l2[t][1]);
      });
    } else if constexpr (AF) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.am[0]), "+v"(a.am[1]), "+v"(a.aa[0]), "+v"(a.aa[1]));
      static_for<HS>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.r[t][0]), "+v"(a.r[t][1]));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = a.r[t][e >> 2][e & 3];
          if (p.a_mul) x = fmaxf(x * a.am[e >> 2][e & 3] + a.aa[e >> 2][e & 3], 0.0f);      // relu(batch-norm(y)), as gemm_f16x3_kernel
          const _Float16 hh = (_Float16)x;
          a.h[t][e] = hh;
          a.l[t][e] = (_Float16)(x - (float)hh);
        }
      });
    } else if constexpr (HS == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.l[0]));
    else if constexpr (HS == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.h[1]), "+v"(a.l[0]), "+v"(a.l[1]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a.h[0]), "+v"(a.h[1]), "+v"(a.h[2]), "+v"(a.h[3]), "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]));
  };
  auto wait_b = [&](FB& b) {
    if constexpr (dbg_nowait) return;
    if constexpr (X1 && WK) {
      static_for<NT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h2[t][0]), "+v"(b.h2[t][1]));
        b.h[t] = join(b.h2[t][0], b.h2[t][1]);
      });
    } else if constexpr (X1) {
      if constexpr (NT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0]));
      else if constexpr (NT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0]), "+v"(b.h[1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0]), "+v"(b.h[1]), "+v"(b.h[2]), "+v"(b.h[3]));
    } else if constexpr (WK) {
      static_for<NT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h2[t][0]), "+v"(b.h2[t][1]), "+v"(b.l2[t][0]), "+v"(b.l2[t][1]));
        b.h[t] = join(b.h2[t][0], b.h2[t][1]);
        b.l[t] = join(b.l2[t][0], b.l2[t][1]);
      });
    } else if constexpr (NT == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0]), "+v"(b.l[0]));
    else if constexpr (NT == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0]), "+v"(b.h[1]), "+v"(b.l[0]), "+v"(b.l[1]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(b.h[0]), "+v"(b.h[1]), "+v"(b.h[2]), "+v"(b.h[3]), "+v"(b.l[0]), "+v"(b.l[1]), "+v"(b.l[2]), "+v"(b.l[3]));
  };
  auto wait_ab = [&](FA& a, FB& b) { wait_a(a); wait_b(b); };
  // MFMA m of a half-step: term-major (lo.hi for every tile, then hi.lo, then hi.hi), so the three terms of one accumulator
  // are HS*NT instructions apart
  auto mma1 = [&](const FA& a, const FB& b, auto mh_c, auto m_c) {
    constexpr int mh = decltype(mh_c)::value, m = decltype(m_c)::value;
    constexpr int term = X1 ? 2 : m / (HS * NT), ii = (m % (HS * NT)) / NT, j = m % NT;
    f32x16& c = acc[HS * mh + ii][j];
    if constexpr (term == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l[ii], b.h[j], c, 0, 0, 0);
    if constexpr (term == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h[ii], b.l[j], c, 0, 0, 0);
    if constexpr (term == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h[ii], b.h[j], c, 0, 0, 0);
  };
  // one half-step: NMMA MFMAs on (a, b) with the NF fillers `fill(0..NF-1)` (fragment reads, then DMA pieces) spread evenly
  // over the gaps behind the MFMAs; sched_barrier(0) pins the order
  auto half_step = [&](const FA& a, const FB& b, auto mh_c, auto nf_c, auto&& fill) {
    constexpr int NF = decltype(nf_c)::value;
    static_for<NMMA>([&](auto m_c) {
      constexpr int m = decltype(m_c)::value;
      mma1(a, b, mh_c, m_c);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CS && m < 2 * HS) {
        // branch-free (cs_one is 0 in the workgroups that do not own the sums): no control flow between the fragment reads in flight
        // and their wait — see the note on wait placement below
        constexpr int ii = m >> 1;
        const half8 v = (m & 1) ? a.l[ii] : a.h[ii];
        float& s_ = bsum[HS * decltype(mh_c)::value + ii];
        s_ = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 0, 1), cs_one, s_, false);
        s_ = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 2, 3), cs_one, s_, false);
        s_ = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 4, 5), cs_one, s_, false);
        s_ = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 6, 7), cs_one, s_, false);
      }
      constexpr int f0 = m * NF / NMMA, f1 = (m + 1) * NF / NMMA;
      static_for<f1 - f0>([&](auto k_c) { fill(std::integral_constant<int, f0 + decltype(k_c)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  constexpr int NP1 = NPW / 2;          // DMA pieces issued in the last half-step of a tile (right after the barrier) ...
  constexpr int NP2 = NPW - NP1;        // ... and in the first half-step of the next one
  constexpr int G0 = RA + RB;           // fillers of a half-step: its fragment reads first, DMA pieces behind them
  auto issue1 = [&](int kt, uint32_t st_off, auto j_c) {
    constexpr int j = decltype(j_c)::value;
    glds16(piece_src(kt, j), dst[j] + st_off);
  };

  // ---- prologue: fill the ring, wait for tile 0, first fragments ---------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nk) issue(s, s * STAGE);
  if (nk >= NS) wait_vmcnt<(NS - 1) * NPW>();             // tile 0 has landed, NS - 1 tiles stay in flight
  else if (NS >= 3 && nk >= 3) wait_vmcnt<(NS >= 3 ? 2 : 0) * NPW>();
  else if (nk >= 2) wait_vmcnt<NPW>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  static_for<RA>([&](auto q_c) { rd_a(fa[0], 0, I0{}, I0{}, q_c); });
  static_for<RB>([&](auto q_c) { rd_b(fb[0], 0, I0{}, q_c); });
  // WAIT PLACEMENT (round 5; the cause of round 4's "two-rank" discrepancy).  The fragment reads are inline asm, so the
  // compiler's own s_waitcnt insertion does not know that their destination registers are in flight until our wait statement, which
  // re-defines them ("+v").  Inside straight-line code the read and its wait share the registers; but a value that is live ACROSS A
  // CONTROL-FLOW EDGE (prologue -> loop, the loop's back edge, loop -> last tile) may be moved by a compiler-inserted copy on that
  // edge — in the column-sum (CS) instantiations hipcc placed eight v_mov_b64 of freshly read fragments ~200 instructions after their
  // ds_read_b64_tr_b16 and BEFORE the wait.  With the LDS pipeline to itself the data had long arrived; with another process's
  // LDS-bound waves on the same CU the copy read the registers' old contents: one 32 x 32 accumulator tile of a weight gradient off
  // by one 16-deep step's contribution, once in ~50 backward passes.  Therefore: every fragment read is waited for in the region
  // that issued it — here, and at the END of tile_body for the next tile's first fragments (dynamically the same place as a wait at
  // the top of the next tile: nothing but the loop branch lies between) — so only waited-for values ever cross an edge.
  // tests/test_abi_and_host.py scans the built library's disassembly for accesses to registers with an LDS read in flight.
  wait_ab(fa[0], fb[0]);

  // One K-tile.  On entry the fragments of half-step 0 (fa[0], fb[0]) are in flight.  LAST: no tile follows.
  // DMA of tile kt + NS - 1 (second half of its pieces) rides in the first half-step, tile kt + NS (first half) in the last one,
  // right behind the barrier that frees its stage.  The prologue issued tiles 0 .. NS-1 completely.
  auto tile_body = [&](int kt, uint32_t cur, uint32_t nxt, uint32_t prv, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    kt_aff = kt;
    const bool dma2 = !(DBG & 2) && kt >= 1 && kt + NS - 1 < nk;     // second half of tile kt + NS - 1 -> stage of tile kt - 1
    const bool dma1 = !(DBG & 2) && !LAST && kt + NS < nk;           // first half of tile kt + NS -> this tile's stage
    auto fill_dma2 = [&](auto m_c) {
      constexpr int m = decltype(m_c)::value;
      if constexpr (m >= G0 && m - G0 < NP2) { if (dma2) issue1(kt + NS - 1, prv, std::integral_constant<int, NP1 + m - G0>{}); }
    };
    auto sync_next = [&]() {
      if constexpr (!(DBG & 4)) {
        if (NS >= 3 && kt + 2 < nk) wait_vmcnt<(NS >= 3 ? NS - 2 : 0) * NPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
    };
    auto fill_next = [&](auto m_c) {       // the next tile's first fragments + the first half of the DMA into the freed stage
      constexpr int m = decltype(m_c)::value;
      if constexpr (!LAST) {
        if constexpr (m < RA) rd_a(fa[0], nxt, I0{}, I0{}, m_c);
        if constexpr (m >= RA && m < G0) rd_b(fb[0], nxt, I0{}, std::integral_constant<int, m - RA>{});
        if constexpr (m >= G0 && m - G0 < NP1) { if (dma1) issue1(kt + NS, cur, std::integral_constant<int, m - G0>{}); }
      }
    };
    if constexpr (NH == 2) {
      // h0 = (s0, first row tiles)   h1 = (s0, last row tiles)   h2 = (s1, first)   h3 = (s1, last)
      // (fa[0], fb[0] were waited for by whoever issued them: the prologue or the previous tile)
      half_step(fa[0], fb[0], I0{}, std::integral_constant<int, G0 + NP2>{}, [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        if constexpr (m < RA) rd_a(fa[1], cur, I0{}, I1{}, m_c);
        fill_dma2(m_c);
      });
      wait_ab(fa[1], fb[0]);
      half_step(fa[1], fb[0], I1{}, std::integral_constant<int, G0>{}, [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        if constexpr (m < RA) rd_a(fa[0], cur, I1{}, I0{}, m_c);
        if constexpr (m >= RA && m < G0) rd_b(fb[1], cur, I1{}, std::integral_constant<int, m - RA>{});
      });
      wait_ab(fa[0], fb[1]);
      half_step(fa[0], fb[1], I0{}, std::integral_constant<int, RA>{}, [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        if constexpr (m < RA) rd_a(fa[1], cur, I1{}, I1{}, m_c);
      });
      // every read of this tile has landed: its stage is free; the next tile must be in LDS before it is read
      wait_ab(fa[1], fb[1]);
      if constexpr (!LAST) sync_next();
      kt_aff = kt + 1;
      half_step(fa[1], fb[1], I1{}, std::integral_constant<int, G0 + NP1>{}, fill_next);
    } else {
      // h0 = s0, h1 = s1 (all row tiles of the wave)
      half_step(fa[0], fb[0], I0{}, std::integral_constant<int, G0 + NP2>{}, [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        if constexpr (m < RA) rd_a(fa[1], cur, I1{}, I0{}, m_c);
        if constexpr (m >= RA && m < G0) rd_b(fb[1], cur, I1{}, std::integral_constant<int, m - RA>{});
        fill_dma2(m_c);
      });
      wait_ab(fa[1], fb[1]);
      if constexpr (!LAST) sync_next();
      kt_aff = kt + 1;
      half_step(fa[1], fb[1], I0{}, std::integral_constant<int, G0 + NP1>{}, fill_next);
    }
    if constexpr (!LAST) wait_ab(fa[0], fb[0]);      // the next tile's first fragments: waited for before the loop edge (see above)
  };

  uint32_t cur = 0, prv = (NS - 1) * STAGE;
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const uint32_t nxt = (cur + STAGE == NS * STAGE) ? 0u : cur + STAGE;
    tile_body(kt, cur, nxt, prv, std::false_type{});
    prv = cur;
    cur = nxt;
  }
  tile_body(nk - 1, cur, 0u, prv, std::true_type{});

  if ((DBG & 1) && acc[0][0][0] != 12345.678f) return;
  if constexpr (CS) {
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float v = (bsum[i] + __shfl_xor(bsum[i], 32)) * p.csum_alpha;
        const int row = m0 + wm * 32 * MT + i * 32 + l31;
        if (lhi == 0 && row < p.M) {
          if (p.split_k == 1) p.csum[row] += v;                                        // one writer per row
          else if (p.csum_ws) p.csum_ws[(size_t)split * p.M + row] = v;                // pl_reduce_kernel adds the chunks in order
          else unsafeAtomicAdd(p.csum + row, v);
        }
      }
    }
  }
  if (p.split_ws && p.split_k > 1) {
    // K split with a workspace: this chunk's alpha * acc goes to its own dense [M, N] slab with 16-byte stores; pl_reduce_kernel
    // adds the slabs in chunk order afterwards (deterministic; fp32 atomics on a shared C measured 1.2 TB/s: every chunk of a
    // tile finishes at the same time and they all hit the same lines)
    GemmP q = p;
    q.C = p.split_ws + (size_t)split * p.M * p.N;
    q.ldc = p.N;
    q.bias = q.scale = q.shift = q.residual = nullptr;
    q.act = PFPP_ACT_NONE;
    q.Chi = q.Clo = nullptr;
    __builtin_amdgcn_s_barrier();
    epilogue_wide<MT, NT>(q, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, lane, 0, 0, lds0 + wave * (32 * NT * 128));
    return;
  }
  if (p.accum) {
    // C += alpha * acc (+ bias / residual from the first K chunk): fp32 atomics; activation and pooling are excluded on the host
    const int row_w = m0 + wm * 32 * MT, col_w = n0 + wn * 32 * NT;
    const float* R = (p.residual && split == 0) ? p.residual + c_off : nullptr;
    const float* bias = (p.bias && split == 0) ? p.bias + v_off : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col_w + j * 32 + l31;
      if (col >= p.N) continue;
      const float b = bias ? bias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (row < p.M) {
            float v = acc[i][j][e] * p.alpha + b;
            if (R) v += R[(int64_t)row * p.ldr + col];
            unsafeAtomicAdd(p.C + c_off + (int64_t)row * p.ldc + col, v);
          }
        }
    }
    return;
  }
  const bool wide_ok = p.pool == 0 && !(DBG & 128) && (p.ldc & 3) == 0 && (p.N & 3) == 0 &&
                       (!p.residual || (p.ldr & 3) == 0) && (p.act != PFPP_ACT_GEGLU || (p.N & 7) == 0);
  if (wide_ok) {
    __builtin_amdgcn_s_barrier();      // every wave is done with the DMA ring: its first bytes become the transposition patches
    epilogue_wide<MT, NT>(p, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, lane, c_off, v_off, lds0 + wave * (32 * NT * 128));
  } else {
    epilogue<MT, NT>(p, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, n0, wn, lane, c_off, v_off);
  }
}

// second pass of a workspace K split: C = [C +] act(sum of the slabs + bias) + residual, four columns per thread
__global__ __launch_bounds__(256) void pl_reduce_kernel(const float* ws, float* C, const float* bias, const float* residual, int M,
                                                        int N, int64_t ldc, int64_t ldr, int splits, int accumulate, int act,
                                                        const float* csum_ws, float* csum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N >> 2;
  if (csum_ws && i < M) {        // the column sums of the A operand, one partial per K chunk
    float s = csum_ws[i];
    for (int k = 1; k < splits; ++k) s += csum_ws[(size_t)k * M + i];
    csum[i] += s;
  }
  if (i >= (int64_t)M * n4) return;
  const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) * 4;
  const size_t slab = (size_t)M * N;
  const float* src = ws + (size_t)row * N + col;
  float4 s = *reinterpret_cast<const float4*>(src);
  for (int k = 1; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(src + k * slab);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + col); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
  s.x = act_apply(s.x, act); s.y = act_apply(s.y, act); s.z = act_apply(s.z, act); s.w = act_apply(s.w, act);
  if (residual) {
    const float4 r = *reinterpret_cast<const float4*>(residual + (int64_t)row * ldr + col);
    s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
  }
  float4* dstp = reinterpret_cast<float4*>(C + (int64_t)row * ldc + col);
  if (accumulate) { const float4 c = *dstp; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
  *dstp = s;
}

// the same sums for up to PFPP_SLAB_GROUP_MAX deferred K splits in one launch (accumulate / store only: no bias, activation, residual)
struct SlabGroupP {
  pfpp_slab_job job[PFPP_SLAB_GROUP_MAX];
  unsigned first_block[PFPP_SLAB_GROUP_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void pl_reduce_group_kernel(const SlabGroupP g) {
  int j = 0;
  while (j + 1 < g.n && blockIdx.x >= g.first_block[j + 1]) ++j;
  const pfpp_slab_job& q = g.job[j];
  const int64_t i = (int64_t)(blockIdx.x - g.first_block[j]) * blockDim.x + threadIdx.x;
  const int M = q.M, N = q.N, n4 = N >> 2, splits = q.splits;
  if (q.csum_ws && i < M) {
    float s = q.csum_ws[i];
    for (int k = 1; k < splits; ++k) s += q.csum_ws[(size_t)k * M + i];
    q.csum[i] += s;
  }
  if (i >= (int64_t)M * n4) return;
  const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) * 4;
  const size_t slab = (size_t)M * N;
  const float* src = q.ws + (size_t)row * N + col;
  float4 s = *reinterpret_cast<const float4*>(src);
  for (int k = 1; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(src + k * slab);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* dstp = reinterpret_cast<float4*>(q.C + (int64_t)row * q.ldc + col);
  if (q.accumulate) { const float4 c = *dstp; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
  *dstp = s;
}

template <int MT, int NT, int WM, int WN, int NS, bool AK, bool WK, int DBG, bool AF = false, bool X1 = false, bool CS = false>
__global__ __launch_bounds__(64 * WM * WN, (MT * NT >= 16 ? 1 : 2)) void gemm_pl_kernel(const GemmP p) {
  pl_body<MT, NT, WM, WN, NS, AK, WK, DBG, AF, X1, CS>(p);
}

// ---- grouped weight gradients: up to PFPP_DW_GROUP_MAX problems dW_j += dY_j^T . X_j (+ db_j += colsum dY_j) over the SAME contraction
// (the token rows of one transformer block's backward) in ONE launch.  The problems' tiles form one concatenated list (row-major
// within a problem: the tiles of a row panel share their dY columns) and XCD x owns a contiguous 1/8 of it (remap_tile), so an XCD
// mostly works on one problem, walks the whole contraction in lock-step over all of its resident tiles and fetches each operand
// column it needs once.  Every tile runs the full contraction in one accumulator chain: no K split, no slabs, no reduction launch;
// the accumulation into the gradient buffer is the in-place residual of the 16-byte-store epilogue (one writer per element).
// The argument carries one complete GemmP per problem (416 bytes each, 3.4 KB of kernel arguments): the workgroup indexes the table in
// the kernarg segment.  (A GemmP assembled in the kernel lived in scratch — 416 bytes per thread, 34 MB of private-segment writes per
// launch: counter WRITE_SIZE 52.8 MB for a 21 MB output in profiles/r05b_*.)
struct DwGroupP {
  GemmP p[PFPP_DW_GROUP_MAX];
  int first_tile[PFPP_DW_GROUP_MAX];
  int n;
};

template <int MT, int NT, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_pl_dwgroup_kernel(const DwGroupP g) {
  const int t = remap_tile(blockIdx.x, gridDim.x);
  int j = 0;
#pragma unroll
  for (int k = 1; k < PFPP_DW_GROUP_MAX; ++k)
    if (k < g.n && t >= g.first_tile[k]) j = k;
  pl_body<MT, NT, WM, WN, NS, true, true, 0, false, false, true, true>(g.p[j], t - g.first_tile[j]);
}

template <int MT, int NT, int WM, int WN, int NS>
int launch_dwgroup(DwGroupP& g, const pfpp_dw_job* jobs, hipStream_t st) {
  using C = Cfg<MT, NT, WM, WN, NS, false>;
  static unsigned long long attr_set = 0;      // one bit per device: the attribute belongs to the function ON a device
  auto kern = gemm_pl_dwgroup_kernel<MT, NT, WM, WN, NS>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set |= 1ull << (dev & 63);
  }
  int tiles = 0;
  for (int j = 0; j < g.n; ++j) {
    g.p[j].tiles_n = (int)((jobs[j].N + C::BN - 1) / C::BN);
    g.first_tile[j] = tiles;
    tiles += (int)((jobs[j].M + C::BM - 1) / C::BM) * g.p[j].tiles_n;
  }
  snprintf(last_kernel, sizeof(last_kernel), "gemm_pl_dwgroup_kernel<%d, %d, %d, %d, %d>", MT, NT, WM, WN, NS);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(C::NTHR), C::SMEM, st, g);
  return pfpp::check_launch("pfpp_gemm_dw_group");
}

template <int MT, int NT, int WM, int WN, int NS, bool AK = false, bool WK = false, int DBG = 0, bool AF = false, bool X1 = false, bool CS = false>
int launch_pl(const GemmP& p0, int batch, hipStream_t st, int group_m, int splits = 1) {
  using C = Cfg<MT, NT, WM, WN, NS, X1>;
#ifdef PFPP_PL_LAB
  if constexpr (DBG == 0 && !AF && !X1) {
    const char* e = getenv("PFPP_GEMM_DBG");
    switch (e ? atoi(e) : 0) {
      case 1: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 1>(p0, batch, st, group_m, splits);
      case 3: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 3>(p0, batch, st, group_m, splits);
      case 7: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 7>(p0, batch, st, group_m, splits);
      case 15: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 15>(p0, batch, st, group_m, splits);
      case 23: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 23>(p0, batch, st, group_m, splits);
      case 128: return launch_pl<MT, NT, WM, WN, NS, AK, WK, 128>(p0, batch, st, group_m, splits);
      default: break;
    }
  }
#endif
  static bool attr_set = false;
  auto kern = gemm_pl_kernel<MT, NT, WM, WN, NS, AK, WK, DBG, AF, X1, CS>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmP p = p0;
  p.tiles_m = (p.M + C::BM - 1) / C::BM;
  p.tiles_n = (p.N + C::BN - 1) / C::BN;
  p.group_m = p.tiles_n > 1 ? group_m : 0;
  const int nk_all = p.K / BK;
  p.split_k = splits < 1 ? 1 : (splits > nk_all ? nk_all : splits);
  const bool slabs = p.split_k > 1 && p.split_ws && batch == 1 && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) &&
                     (int64_t)p.split_k * p.M * (p.N + 1) * (int64_t)sizeof(float) <= p_ws_bytes;
  if (!slabs) p.split_ws = nullptr;
  if constexpr (!CS) p.csum = nullptr;
  // partial column sums of the chunks live behind the C slabs (the capacity check of the caller includes them)
  p.csum_ws = (CS && slabs && p.csum) ? p.split_ws + (size_t)p.split_k * p.M * p.N : nullptr;
  if (p.split_k > 1 && !slabs) {
    if (p.act != PFPP_ACT_NONE || !p.accum) { p.split_k = 1; }     // atomics need an accumulating, activation-free epilogue
  }
  p.k_chunk = 0;
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.split_k), 1, (unsigned)batch);
  snprintf(last_kernel, sizeof(last_kernel), "gemm_pl_kernel<%d, %d, %d, %d, %d, %s, %s, %d, %s, %s, %s>%s", MT, NT, WM, WN, NS, AK ? "true" : "false",
           WK ? "true" : "false", DBG, AF ? "true" : "false", X1 ? "true" : "false", CS ? "true" : "false", slabs ? "+pl_reduce_kernel" : "");
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), C::SMEM + (AF ? 2048 : 0), st, p);
  if (p.defer) {
    pfpp_slab_job& q = *p.defer;
    q = pfpp_slab_job{};
    if (slabs) {
      q.ws = p.split_ws; q.C = p.C; q.csum_ws = p.csum_ws; q.csum = p.csum;
      q.M = p.M; q.N = p.N; q.ldc = p.ldc; q.splits = p.split_k; q.accumulate = p.accum;
    }
  } else if (slabs) {
    const int64_t n4 = (int64_t)p.M * (p.N >> 2);
    hipLaunchKernelGGL(pl_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p.split_ws, p.C, p.bias, p.residual, p.M,
                       p.N, p.ldc, p.ldr, p.split_k, p.accum, p.act, p.csum_ws, p.csum);
  }
  return pfpp::check_launch("pfpp_gemm");
}


}  // namespace pl

// variant: 0 = pick by shape, 1 = 256x256 (8 waves of 128x64, 2 stages), 2 = 256x128 (8 waves, 3 stages), 3 = 128x128 (4 waves,
// 2 stages, two workgroups per CU), 6 = 128x64 (3 stages): small tiles for narrow outputs
// single-pass fp16 (hi planes only, one MFMA per product): the perf mode of BASELINE configs[4] — no fp32-grade guarantee
static int launch_variant_x1(const GemmP& p, int batch, hipStream_t st, int group_m, int variant, int splits) {
  switch (variant) {
    case 1: return pl::launch_pl<4, 2, 2, 4, 2, false, false, 0, false, true>(p, batch, st, group_m, splits);
    case 2: return pl::launch_pl<2, 2, 4, 2, 3, false, false, 0, false, true>(p, batch, st, group_m, splits);
    case 6: return pl::launch_pl<2, 1, 2, 2, 3, false, false, 0, false, true>(p, batch, st, group_m, splits);
    default: return pl::launch_pl<2, 2, 2, 2, 2, false, false, 0, false, true>(p, batch, st, group_m, splits);
  }
}

template <bool AK, bool WK>
static int launch_variant(const GemmP& p, int batch, hipStream_t st, int group_m, int variant, int splits) {
  if constexpr (AK && WK) {
    if (p.csum) {      // dW = dY^T . X with the bias gradient (column sums of dY) computed on the way
      switch (variant) {
        case 1: case 2: return pl::launch_pl<2, 2, 4, 2, 3, AK, WK, 0, false, false, true>(p, batch, st, group_m, splits);
        case 6: return pl::launch_pl<2, 1, 2, 2, 3, AK, WK, 0, false, false, true>(p, batch, st, group_m, splits);
        default: return pl::launch_pl<2, 2, 2, 2, 2, AK, WK, 0, false, false, true>(p, batch, st, group_m, splits);
      }
    }
  }
  switch (variant) {
    case 1:
      if constexpr (!AK && !WK) return pl::launch_pl<4, 2, 2, 4, 2, AK, WK>(p, batch, st, group_m, splits);
      [[fallthrough]];
    case 2: return pl::launch_pl<2, 2, 4, 2, 3, AK, WK>(p, batch, st, group_m, splits);
    case 6: return pl::launch_pl<2, 1, 2, 2, 3, AK, WK>(p, batch, st, group_m, splits);
    case 9:       // 64 x 64 (4 waves of 32 x 32, three stages): grids of a few dozen workgroups — twice the workgroups per output, half
                  // the bytes per K-tile of each: a single puzzle's 200-row GEMMs are bound by the per-workgroup DMA rate
      if constexpr (!AK && !WK) return pl::launch_pl<1, 1, 2, 2, 3, AK, WK>(p, batch, st, group_m, splits);
      return pl::launch_pl<2, 1, 2, 2, 3, AK, WK>(p, batch, st, group_m, splits);
    case 10:      // 128 x 32, two waves: twice the workgroups along N, each streaming half the weight columns
      if constexpr (!AK && !WK) return pl::launch_pl<2, 1, 2, 1, 3, AK, WK>(p, batch, st, group_m, splits);
      return pl::launch_pl<2, 1, 2, 2, 3, AK, WK>(p, batch, st, group_m, splits);
    case 11:      // 64 x 32, two waves
      if constexpr (!AK && !WK) return pl::launch_pl<1, 1, 2, 1, 3, AK, WK>(p, batch, st, group_m, splits);
      return pl::launch_pl<2, 1, 2, 2, 3, AK, WK>(p, batch, st, group_m, splits);
    case 12:      // 64 x 64 as two waves of 32 x 64: the GEGLU form (value | gate column tiles in one wave) of the above
      if constexpr (!AK && !WK) return pl::launch_pl<1, 2, 2, 1, 3, AK, WK>(p, batch, st, group_m, splits);
      return pl::launch_pl<2, 2, 2, 2, 2, AK, WK>(p, batch, st, group_m, splits);
    default: return pl::launch_pl<2, 2, 2, 2, 2, AK, WK>(p, batch, st, group_m, splits);
  }
}

// fp32 A read in place (LDS-DMA of fp32 rows, converted when a wave fetches its fragments) with the train-mode BatchNorm + ReLU of
// the previous set-abstraction layer fused in: the [1.26 M x 64..256] encoder GEMMs, bound by HBM, not by the conversions
int launch_f16x3_planes_af32(const GemmP& p0, int batch, hipStream_t st, int group_m) {
  GemmP p = p0;
  if (getenv("PFPP_DIAG_NO_STATS")) p.stats = nullptr;       // diagnostic only: what do the statistics atomics cost?
  if (getenv("PFPP_DIAG_NO_POOL")) { p.pool = 0; p.Cmin = nullptr; }
  // measured on the train-mode set-abstraction shapes (tools/diag/gemm_calls.py): every tile / stage choice lands within 10 % — these
  // launches are bound by the per-workgroup fixed costs of a 2..8 K-tile contraction (cold-HBM prologue, epilogue) and by HBM
  // itself (3.6 TB/s on the layers that write their activation), not by the K loop.  Two or three co-resident workgroups per CU
  // hide a little more of the prologue than one large one.
  static const int afv = getenv("PFPP_GEMM_AF32_VARIANT") ? atoi(getenv("PFPP_GEMM_AF32_VARIANT")) : 0;
  if (afv == 4) return pl::launch_pl<2, 2, 4, 2, 3, false, false, 0, true>(p, batch, st, group_m, 1);       // 256 x 128, one per CU
  if (p.N <= 64) return pl::launch_pl<2, 1, 2, 2, 2, false, false, 0, true>(p, batch, st, group_m, 1);      // 128 x 64, three per CU
  return pl::launch_pl<2, 2, 2, 2, 2, false, false, 0, true>(p, batch, st, group_m, 1);                      // 128 x 128, two per CU
}

// Tile and K split from a small cost model fitted to tools/gemm_lab measurements on 3,850-row shapes: a workgroup's main
// loop is bound by operand delivery (~45 GB/s of L2 -> LDS DMA per CU, shared by co-resident workgroups) or by its MFMAs
// (32 cycles each, ~75 % sustained); the epilogue streams C once (3.5 TB/s) — or, for a K split, writes and re-reads one
// slab per chunk plus a second launch (6 us); atomics onto a shared C run at 1.2 TB/s.  fix_variant / fix_splits != 0 pin a choice.
static void pl_choose(int M, int N, int K, int fix_variant, int fix_splits, bool have_ws, int64_t ws_bytes, bool accumulate, bool no_v6,
                      int* out_v, int* out_s, bool allow_v1 = false, bool tn_form = false) {
  const int nk = K / 32;
  const int cand_v[4] = {2, 3, 6, 1};
  const int cand_s[9] = {1, 2, 3, 4, 6, 8, 10, 12, 16};
  double best = 1e30;
  int best_v = 3, best_s = 1;
  for (int vi = 0; vi < 4; ++vi) {
    const int v = cand_v[vi];
    if (fix_variant != 0 && v != fix_variant) continue;
    if (no_v6 && v == 6) continue;
    if (v == 1 && !allow_v1 && fix_variant != 1) continue;          // 256 x 256: row-major operands only
    const int bm = (v == 2 || v == 1) ? 256 : 128, bn = v == 6 ? 64 : (v == 1 ? 256 : 128);
    const double tiles = (double)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    for (int si = 0; si < 9; ++si) {
      const int sp = cand_s[si];
      if (fix_splits != 0 && sp != fix_splits) continue;
      if (sp > 1 && (sp > nk / 2 || !(have_ws || accumulate))) continue;
      const bool slabs = sp > 1 && have_ws && (double)sp * M * (N + 1.0) * 4.0 <= (double)ws_bytes;
      if (sp > 1 && !slabs && !accumulate) continue;
      const double wgs = tiles * sp;
      // co-resident workgroups share the CU's bandwidth; a grid that is not a multiple of the CU count leaves some CUs with one
      // workgroup more than the others (measured: between the fractional and the rounded-up count)
      const double frac = wgs / 256.0;
      const double rounds = wgs <= 256.0 ? 1.0 : 0.5 * (frac + ceil(frac));
      const double kc = (double)K / sp;
      const double t_bw = rounds * kc * (bm + bn) * 4.0 / 45e9;
      const double t_mma = rounds * (bm / 32.0) * (bn / 32.0) * (kc / 16.0) * 3.0 * 32.0 / 4.0 / 2.1e9 / 0.75;
      const double cbytes = (double)M * N * 4.0;
      double t_epi;
      if (slabs) t_epi = (2.0 * sp * cbytes + (accumulate ? 2.0 : 1.0) * cbytes) / 3.5e12 + 6e-6;      // + the reduction launch (6.6 us measured)
      else if (accumulate) t_epi = sp * cbytes / 1.2e12;
      else t_epi = cbytes / 3.5e12;
      double t_loop = t_bw > t_mma ? t_bw : t_mma;
      // dW form (both operands k-major: two transposing LDS reads per MFMA operand): the two-stage 128 x 128 tile measures 20-25 %
      // above its delivery bound there (1536x512x3850: 40.3 us vs 32.1 us for the three-stage 128 x 64 tile at the same split)
      // (off by default: stand-alone it is right — serial iteration 9.55 -> 9.33 ms — but it moves the weight gradients to the
      // 144 KB-LDS 256 x 128 tile, which then owns whole CUs next to the main chain: overlapped iteration 8.30 -> 8.57 ms)
      static const bool tn_pen = getenv("PFPP_PL_TN_PENALTY") && atoi(getenv("PFPP_PL_TN_PENALTY")) == 1;
      if (tn_form && v == 3 && tn_pen) t_loop *= 1.25;
      const double t = t_loop + t_epi + 4e-6;
      if (t < best) { best = t; best_v = v; best_s = sp; }
    }
  }
  *out_v = best_v;
  *out_s = best_s;
  static const bool dbg = getenv("PFPP_GEMM_CHOICE_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "pl_choose M%d N%d K%d -> variant %d splits %d (model %.1f us)\n", M, N, K, best_v, best_s, best * 1e6);
}

int launch_f16x3_planes(const GemmP& p, int batch, hipStream_t st, int group_m, int variant) {
  pl::p_ws_bytes = p.split_ws ? p.ws_bytes : 0;
  // few output tiles and an epilogue the slab reduction can express (bias / activation / residual, fp32 out): split the
  // contraction over the idle CUs — the 100..500-row GEMMs of a single puzzle's DDPM step (16 tiles x 16 K-tiles otherwise)
  if (variant == 0 && batch == 1 && p.split_ws && !p.x1 && p.pool == 0 && !p.stats && !p.scale && !p.Chi && !p.Cmin &&
      p.act != PFPP_ACT_GEGLU && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) &&
      (int64_t)((p.M + 127) / 128) * ((p.N + 63) / 64) <= 96) {
    // measured on the single-puzzle loop (bench.aggl_puzzles_per_s, one puzzle in flight): 5.28 -> 4.86 puzzles/s — the slab epilogues
    // and the second launch cost more than the shorter K loops save; off unless asked for
    static const bool on = getenv("PFPP_GEMM_SMALL_SPLIT") && atoi(getenv("PFPP_GEMM_SMALL_SPLIT")) == 1;
    if (on) {
      int v = 3, sp = 1;
      pl_choose(p.M, p.N, p.K, 0, 0, true, pl::p_ws_bytes, false, false, &v, &sp);
      if (sp > 1) return launch_variant<false, false>(p, batch, st, group_m, v, sp);
    }
  }
  static const int chooser = getenv("PFPP_GEMM_CHOOSER") ? atoi(getenv("PFPP_GEMM_CHOOSER")) : 1;      // 0: tile-count rule; 2: no 256 x 256
  if (chooser && variant == 0 && batch == 1 && !p.x1 && p.pool == 0 && !p.stats && p.M <= 16384) {
    // token-sized GEMMs of the transformer: the same cost model as pfpp_gemm_planes (tile only; no K split on this path)
    int v = 3, sp = 1;
    pl_choose(p.M, p.N, p.K, 0, 1, false, 0, false, p.act == PFPP_ACT_GEGLU, &v, &sp, chooser != 2);
    variant = v;
  }
  if (variant == 0) {
    const int64_t t256 = ((int64_t)(p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
    const int64_t t21 = ((int64_t)(p.M + 255) / 256) * ((p.N + 127) / 128) * batch;
    const int64_t t11 = ((int64_t)(p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    variant = t256 >= 512 ? 1 : (t21 >= 160 ? 2 : (t11 >= 200 ? 3 : 6));
    // (a six-stage ring for grids of a few workgroups was tried for the single-puzzle loop: 5.50 -> 5.55 puzzles/s, not kept)
  }
  // a single puzzle's GEMMs (200 rows): each workgroup streams its weight columns cold (57.6 M parameters do not stay in L2 between
  // steps) at the per-workgroup DMA rate, so narrower tiles = more workgroups pulling in parallel: one puzzle in flight 5.1-5.3 ->
  // 5.4-5.5 puzzles/s with the 64 x 32 tile (9 = 64 x 64: no gain, same columns per workgroup; 0 = keep 128 x 64)
  static const int tiny = getenv("PFPP_GEMM_TINY_TILE") ? atoi(getenv("PFPP_GEMM_TINY_TILE")) : 11;
  if (tiny && variant == 6 && batch == 1 && !p.x1 && p.pool == 0 && p.act != PFPP_ACT_GEGLU &&
      (int64_t)((p.M + 127) / 128) * ((p.N + 63) / 64) <= 48)
    variant = tiny;
  static const int tiny_g = getenv("PFPP_GEMM_TINY_GEGLU") ? atoi(getenv("PFPP_GEMM_TINY_GEGLU")) : 0;
  if (tiny_g && batch == 1 && !p.x1 && p.pool == 0 && p.act == PFPP_ACT_GEGLU && (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) <= 96)
    variant = 12;
  // GEGLU gates pairs of column tiles (two per wave at least); the pool = 64 epilogue needs two row tiles per wave
  if (variant == 6 && p.act == PFPP_ACT_GEGLU) variant = 3;
  if (variant == 1 && p.pool == 64) variant = 2;
  if (p.x1) return launch_variant_x1(p, batch, st, group_m, variant, 1);
  return launch_variant<false, false>(p, batch, st, group_m, variant, 1);
}

}  // namespace pfpp_gemm_detail

using namespace pfpp_gemm_detail;

extern "C" const char* pfpp_last_gemm_kernel(void) { return pl::last_kernel; }

extern "C" int pfpp_slab_reduce_group(const pfpp_slab_job* jobs, int32_t n_jobs, pfpp_stream_t stream) {
  PFPP_REQUIRE(jobs && n_jobs >= 0 && n_jobs <= PFPP_SLAB_GROUP_MAX, "0..PFPP_SLAB_GROUP_MAX jobs");
  pl::SlabGroupP g;
  memset(&g, 0, sizeof(g));
  unsigned blocks = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const pfpp_slab_job& q = jobs[j];
    if (q.splits == 0) continue;
    PFPP_REQUIRE(q.ws && q.C && q.M > 0 && q.N > 0 && q.N % 4 == 0 && q.ldc % 4 == 0 && q.splits >= 2 && (!q.csum_ws || q.csum) &&
                 pfpp::aligned16(q.ws) && pfpp::aligned16(q.C), "slab job");
    g.job[g.n] = q;
    g.first_block[g.n] = blocks;
    blocks += (unsigned)(((int64_t)q.M * (q.N >> 2) + 255) / 256);
    ++g.n;
  }
  g.first_block[g.n] = blocks;
  if (g.n == 0) return PFPP_OK;
  hipLaunchKernelGGL(pl::pl_reduce_group_kernel, dim3(blocks), dim3(256), 0, pfpp::as_stream(stream), g);
  return pfpp::check_launch("pfpp_slab_reduce_group");
}

extern "C" int pfpp_gemm_dw_group(const pfpp_dw_job* jobs, int32_t n_jobs, int64_t K, int32_t variant, pfpp_stream_t stream) {
  PFPP_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= PFPP_DW_GROUP_MAX, "1 .. PFPP_DW_GROUP_MAX jobs");
  PFPP_REQUIRE(K > 0 && K < (1ll << 31), "contraction length");
  pl::DwGroupP g;
  memset(&g, 0, sizeof(g));
  g.n = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const pfpp_dw_job& q = jobs[j];
    PFPP_REQUIRE(q.dy.hi && q.dy.lo && q.x.hi && q.x.lo && q.gw, "null pointer");
    PFPP_REQUIRE(q.M >= 8 && q.N >= 8 && q.M % 8 == 0 && q.N % 8 == 0 && q.M < (1ll << 31) && q.N < (1ll << 31), "M, N: multiples of 8");
    PFPP_REQUIRE(pfpp::aligned16(q.dy.hi) && pfpp::aligned16(q.dy.lo) && pfpp::aligned16(q.x.hi) && pfpp::aligned16(q.x.lo) &&
                 pfpp::aligned16(q.gw), "16-byte aligned operands");
    PFPP_REQUIRE(q.dy.scale > 0.0f && q.x.scale > 0.0f, "plane scales");
    GemmP& p = g.p[j];                       // (zeroed above: every optional pointer null, pool / act / stats off)
    p.Ahi = q.dy.hi; p.Alo = q.dy.lo; p.Whi = q.x.hi; p.Wlo = q.x.lo;
    p.C = q.gw; p.residual = q.gw;           // C += alpha * acc: the in-place residual of the wide epilogue
    p.M = (int)q.M; p.N = (int)q.N; p.K = (int)((K + 31) / 32 * 32); p.k_valid = (int)K;
    p.lda = q.M; p.ldw = q.N; p.ldc = q.N; p.ldr = q.N;
    p.act = PFPP_ACT_NONE; p.zdiv = 1; p.split_k = 1; p.stats_copies = 1;
    p.alpha = 1.0f / (q.dy.scale * q.x.scale);
    p.csum = q.gb; p.csum_alpha = 1.0f / q.dy.scale;
  }
  hipStream_t st = pfpp::as_stream(stream);
  const int env_v = getenv("PFPP_DW_GROUP_VARIANT") ? atoi(getenv("PFPP_DW_GROUP_VARIANT")) : 0;      // (read per call, like PFPP_TRAIN_DW_GROUP*: the tests switch it)
  // default: the 256 x 128 tile (8 waves, one workgroup per CU).  Measured on a block's six problems at 3,850 tokens
  // (profiles/r05a_lab_dw_group_bench.txt): 166 us against 195-213 us for the 128 x 64 / 128 x 128 tiles (and 256 us for six separate
  // launches + their slab reductions) although its 160 tiles leave 96 CUs without one; in the overlapped training iteration 6.06 ms
  // against 6.28 (small tiles) and 6.23 (separate launches) — profiles/r05a_ab_dwgroup.txt
  const int v = variant ? variant : (env_v ? env_v : 2);
  switch (v) {
    case 6: return pl::launch_dwgroup<2, 1, 2, 2, 3>(g, jobs, st);       // 128 x 64, three stages (72 KB: two workgroups per CU)
    case 7: return pl::launch_dwgroup<2, 1, 2, 2, 2>(g, jobs, st);       // 128 x 64, two stages (48 KB: three per CU)
    case 2: return pl::launch_dwgroup<2, 2, 4, 2, 3>(g, jobs, st);       // 256 x 128, 8 waves
    default: return pl::launch_dwgroup<2, 2, 2, 2, 2>(g, jobs, st);      // 128 x 128, two stages (64 KB: two per CU)
  }
}

extern "C" int pfpp_gemm_planes(const pfpp_gemm_planes_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a && a->a_hi && a->a_lo && a->w_hi && a->w_lo && a->C, "null pointer");
  PFPP_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "sizes must be positive");
  PFPP_REQUIRE(a->K % 32 == 0 || (a->a_kmajor && a->w_kmajor), "K % 32 != 0 is only possible with both operands k-major");
  PFPP_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "sizes exceed int32");
  PFPP_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && pfpp::aligned16(a->a_hi) && pfpp::aligned16(a->a_lo) &&
               pfpp::aligned16(a->w_hi) && pfpp::aligned16(a->w_lo), "planes: 16-byte aligned, leading dimensions % 8 == 0");
  PFPP_REQUIRE(a->a_kmajor ? (a->M % 8 == 0 && a->lda >= a->M) : a->lda >= a->K, "A: lda too small (k-major: M % 8 == 0)");
  PFPP_REQUIRE(a->w_kmajor ? (a->N % 8 == 0 && a->ldw >= a->N) : a->ldw >= a->K, "W: ldw too small (k-major: N % 8 == 0)");
  PFPP_SUPPORTED(!(a->a_kmajor && !a->w_kmajor), "k-major A with a row-major W");
  PFPP_REQUIRE(!a->residual || a->ldr > 0, "residual without ldr");
  PFPP_REQUIRE(a->act >= PFPP_ACT_NONE && a->act <= PFPP_ACT_GELU, "activation");
  PFPP_REQUIRE(a->splits >= 0 && (a->splits <= 1 || a->accumulate || a->ws), "split-K needs a workspace or accumulate");
  PFPP_REQUIRE(!a->accumulate || a->act == PFPP_ACT_NONE, "accumulate excludes an activation");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.C = a->C; p.Ahi = a->a_hi; p.Alo = a->a_lo; p.Whi = a->w_hi; p.Wlo = a->w_lo;
  p.bias = a->bias; p.residual = a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)((a->K + 31) / 32 * 32);
  p.k_valid = (int)a->K;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
  p.act = a->act; p.zdiv = 1; p.alpha = a->alpha; p.accum = a->accumulate ? 1 : 0;
  p.split_ws = a->ws;
  p.ws_bytes = a->ws ? a->ws_bytes : 0;
  pl::p_ws_bytes = p.ws_bytes;
  PFPP_SUPPORTED(!a->colsum || (a->a_kmajor && a->w_kmajor && !a->single_pass), "colsum rides with the k-major pair (dW = dY^T . X) only");
  p.csum = a->colsum; p.csum_alpha = a->colsum_alpha;
  PFPP_SUPPORTED(!a->defer || (!a->bias && !a->residual && a->act == PFPP_ACT_NONE && a->ws), "defer: plain (accumulating) outputs with a workspace only");
  p.defer = a->defer;
  hipStream_t st = pfpp::as_stream(stream);
  int variant = a->variant, splits = a->splits;
  const int nk = p.K / 32;
  if (variant == 0 || splits == 0) {
    int best_v = 3, best_s = 1;
    pl_choose(p.M, p.N, p.K, a->variant, a->splits, a->ws != nullptr, a->ws ? a->ws_bytes : 0, a->accumulate != 0, false, &best_v, &best_s,
              !a->a_kmajor && !a->w_kmajor && !a->single_pass, a->a_kmajor && a->w_kmajor);
    if (variant == 0) variant = best_v;
    if (splits == 0) splits = best_s;
  }
  static const int gm_env = getenv("PFPP_GEMM_GROUP_M") ? atoi(getenv("PFPP_GEMM_GROUP_M")) : 8;
  const int gm = gm_env;
  if (a->single_pass) {
    PFPP_SUPPORTED(!a->a_kmajor && !a->w_kmajor, "single-pass fp16 with k-major operands");
    p.x1 = 1;
    return launch_variant_x1(p, 1, st, gm, variant, splits);
  }
  if (a->a_kmajor) return launch_variant<true, true>(p, 1, st, gm, variant, splits);
  if (a->w_kmajor) return launch_variant<false, true>(p, 1, st, gm, variant, splits);
  return launch_variant<false, false>(p, 1, st, gm, variant, splits);
}
