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
// Fragment reads are inline asm (hipcc drains vmcnt(0) in front of every ds_read it can see while an LDS-DMA is
// in flight); the waits name every destination register, which is what orders the MFMAs behind them.
// One __shared__ object only (a second one makes hipcc drain the DMA queue at every step, guide §5).
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

namespace pfpp_gemm_detail {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace pl {

constexpr int BK = 32;

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
template <int MT, int NT, int WM, int WN, int NS>
struct Cfg {
  static constexpr int HS = (MT == 4 && NT == 2) ? 2 : MT;
  static constexpr int WPS = (MT * NT >= 16) ? 1 : 2;     // waves per SIMD the register budget allows (512 / 256 registers)
  static constexpr int NW = WM * WN;
  static constexpr int NTHR = 64 * NW;
  static constexpr int BM = 32 * MT * WM;
  static constexpr int BN = 32 * NT * WN;
  static constexpr int PLANE_A = BM * 64;                 // bytes: BM rows of 32 halfs
  static constexpr int PLANE_W = BN * 64;
  static constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_W;
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
  float sc[NT], sh[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = col_w + j * 32 + l31;
    sc[j] = 1.0f; sh[j] = 0.0f;
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
#pragma unroll
    for (int k = 0; k < 32 / RPI; ++k) {
      const int r = rrow + k * RPI;
      float4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(patch + r * ROWB + rcol * 4) : "memory");
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
}

// One workgroup = one output tile.  NS-stage DMA ring; see the header for the schedule.
template <int MT, int NT, int WM, int WN, int NS, int DBG>
__device__ __forceinline__ void pl_body(const GemmP& p) {
  using C = Cfg<MT, NT, WM, WN, NS>;
  constexpr int BM = C::BM, BN = C::BN, NPW = C::NPW, STAGE = C::STAGE;
  constexpr int HS = C::HS;
  constexpr int NH = MT / HS;         // half-steps per 16-deep step
  constexpr int NMMA = 3 * HS * NT;   // MFMAs per half-step
  constexpr int RA = 2 * HS, RB = 2 * NT;   // fragment reads of a half-step's A tiles / a step's W tiles
  extern __shared__ __align__(1024) char pl_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int tile = remap_tile(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(p, tile, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const int64_t a_offz = z0 * p.sA0 + z1 * p.sA1;
  const int64_t w_offz = z0 * p.sW0 + z1 * p.sW1;
  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)pl_smem;

  // ---- DMA sources: piece q = wave + NW*j of a stage = LDS bytes [q KiB, (q+1) KiB) = 16 rows of one plane --------
  // lane i of a piece lands at byte q*1024 + i*16: row (i >> 2) of the piece, physical chunk i & 3, which holds the
  // row's logical 16-byte chunk (i & 3) ^ ((row >> 2) & 3)
  const char* src[NPW];
  uint32_t dst[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int q = wave + C::NW * j;
    const int o = q * 1024;
    const bool is_a = o < 2 * C::PLANE_A;
    const int o2 = is_a ? o : o - 2 * C::PLANE_A;
    const int psz = is_a ? C::PLANE_A : C::PLANE_W;
    const bool lo = o2 >= psz;
    const int row = ((lo ? o2 - psz : o2) >> 6) + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const _Float16* base = reinterpret_cast<const _Float16*>(is_a ? (lo ? p.Alo : p.Ahi) : (lo ? p.Wlo : p.Whi));
    const int64_t eoff = is_a ? a_offz + (int64_t)min(m0 + row, p.M - 1) * p.lda
                              : w_offz + (int64_t)min(n0 + row, p.N - 1) * p.ldw;
    src[j] = reinterpret_cast<const char*>(base + eoff + chunk * 8);
    dst[j] = lds0 + o;
  }
  auto issue = [&](int kt, uint32_t st_off) {
    const int kb = kt * (BK * 2);
#pragma unroll
    for (int j = 0; j < NPW; ++j) glds16(src[j] + kb, dst[j] + st_off);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // ---- fragment addressing: lane (l31, lhi) reads row l31 of a 32-row tile, logical chunk 2*s + lhi --------------------
  const int sw = (l31 >> 2) & 3;
  uint32_t a_ad[2], w_ad[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_ad[s] = lds0 + (wm * 32 * MT + l31) * 64 + (((2 * s + lhi) ^ sw) << 4);
    w_ad[s] = lds0 + 2 * C::PLANE_A + (wn * 32 * NT + l31) * 64 + (((2 * s + lhi) ^ sw) << 4);
  }
  struct FA { half8 h[HS], l[HS]; };
  struct FB { half8 h[NT], l[NT]; };
  FA fa[2];
  FB fb[2];
  // DBG (lab builds, -DPFPP_PL_LAB): 1 no epilogue, 2 no DMA after the prologue, 4 no barrier / DMA wait, 8 no fragment reads,
  // 16 no fragment waits
  constexpr bool dbg_noread = DBG & 8, dbg_nowait = DBG & 16;
  // Fragment reads one at a time (q < HS: hi plane of row tile q, then the lo planes), each in its own gap between two MFMAs
  auto rd_a = [&](FA& f, uint32_t st, auto s_c, auto mh_c, auto q_c) {
    constexpr int s = decltype(s_c)::value, mh = decltype(mh_c)::value, q = decltype(q_c)::value;
    if constexpr (dbg_noread) return;
    const uint32_t ad = a_ad[s] + st;
    constexpr int t = q % HS;
    if constexpr (q < HS) f.h[t] = lds_rd<(HS * mh + t) * 2048>(ad);
    else f.l[t] = lds_rd<C::PLANE_A + (HS * mh + t) * 2048>(ad);
  };
  auto rd_b = [&](FB& f, uint32_t st, auto s_c, auto q_c) {
    constexpr int s = decltype(s_c)::value, q = decltype(q_c)::value;
    if constexpr (dbg_noread) return;
    const uint32_t ad = w_ad[s] + st;
    constexpr int t = q % NT;
    if constexpr (q < NT) f.h[t] = lds_rd<t * 2048>(ad);
    else f.l[t] = lds_rd<C::PLANE_W + t * 2048>(ad);
  };
  // all outstanding LDS reads have landed; names the fragments the following MFMAs consume
  auto wait_a = [&](FA& a) {
    if constexpr (dbg_nowait) return;
    if constexpr (HS == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.l[0]));
    else if constexpr (HS == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.h[0]), "+v"(a.h[1]), "+v"(a.l[0]), "+v"(a.l[1]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a.h[0]), "+v"(a.h[1]), "+v"(a.h[2]), "+v"(a.h[3]), "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]));
  };
  auto wait_b = [&](FB& b) {
    if constexpr (dbg_nowait) return;
    if constexpr (NT == 1)
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
    constexpr int term = m / (HS * NT), ii = (m % (HS * NT)) / NT, j = m % NT;
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
      constexpr int f0 = m * NF / NMMA, f1 = (m + 1) * NF / NMMA;
      static_for<f1 - f0>([&](auto k_c) { fill(std::integral_constant<int, f0 + decltype(k_c)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  const int nk = p.K / BK;
  constexpr int NP1 = NPW / 2;          // DMA pieces issued in the last half-step of a tile (right after the barrier) ...
  constexpr int NP2 = NPW - NP1;        // ... and in the first half-step of the next one
  constexpr int G0 = RA + RB;           // fillers of a half-step: its fragment reads first, DMA pieces behind them
  auto issue1 = [&](int kt, uint32_t st_off, auto j_c) {
    constexpr int j = decltype(j_c)::value;
    glds16(src[j] + kt * (BK * 2), dst[j] + st_off);
  };

  // ---- prologue: fill the ring, wait for tile 0, first fragments ---------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nk) issue(s, s * STAGE);
  if (NS >= 3 && nk >= 3) wait_vmcnt<(NS >= 3 ? 2 : 0) * NPW>();
  else if (nk >= 2) wait_vmcnt<NPW>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  static_for<RA>([&](auto q_c) { rd_a(fa[0], 0, I0{}, I0{}, q_c); });
  static_for<RB>([&](auto q_c) { rd_b(fb[0], 0, I0{}, q_c); });

  // One K-tile.  On entry the fragments of half-step 0 (fa[0], fb[0]) are in flight.  LAST: no tile follows.
  // DMA of tile kt + NS - 1 (second half of its pieces) rides in the first half-step, tile kt + NS (first half) in the last one,
  // right behind the barrier that frees its stage.  The prologue issued tiles 0 .. NS-1 completely.
  auto tile_body = [&](int kt, uint32_t cur, uint32_t nxt, uint32_t prv, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
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
      wait_ab(fa[0], fb[0]);
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
      half_step(fa[1], fb[1], I1{}, std::integral_constant<int, G0 + NP1>{}, fill_next);
    } else {
      // h0 = s0, h1 = s1 (all row tiles of the wave)
      wait_ab(fa[0], fb[0]);
      half_step(fa[0], fb[0], I0{}, std::integral_constant<int, G0 + NP2>{}, [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        if constexpr (m < RA) rd_a(fa[1], cur, I1{}, I0{}, m_c);
        if constexpr (m >= RA && m < G0) rd_b(fb[1], cur, I1{}, std::integral_constant<int, m - RA>{});
        fill_dma2(m_c);
      });
      wait_ab(fa[1], fb[1]);
      if constexpr (!LAST) sync_next();
      half_step(fa[1], fb[1], I0{}, std::integral_constant<int, G0 + NP1>{}, fill_next);
    }
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
  const bool wide_ok = p.pool == 0 && !p.stats && !(DBG & 128) && (p.ldc & 3) == 0 && (p.N & 3) == 0 &&
                       (!p.residual || (p.ldr & 3) == 0) && (p.act != PFPP_ACT_GEGLU || (p.N & 7) == 0);
  if (wide_ok) {
    __builtin_amdgcn_s_barrier();      // every wave is done with the DMA ring: its first bytes become the transposition patches
    epilogue_wide<MT, NT>(p, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, lane, c_off, v_off, lds0 + wave * (32 * NT * 128));
  } else {
    epilogue<MT, NT>(p, acc, m0 + wm * 32 * MT, n0 + wn * 32 * NT, n0, wn, lane, c_off, v_off);
  }
}

template <int MT, int NT, int WM, int WN, int NS, int DBG>
__global__ __launch_bounds__(64 * WM * WN, (MT * NT >= 16 ? 1 : 2)) void gemm_pl_kernel(const GemmP p) {
  pl_body<MT, NT, WM, WN, NS, DBG>(p);
}

template <int MT, int NT, int WM, int WN, int NS, int DBG = 0>
int launch_pl(const GemmP& p0, int batch, hipStream_t st, int group_m) {
  using C = Cfg<MT, NT, WM, WN, NS>;
#ifdef PFPP_PL_LAB
  if constexpr (DBG == 0) {
    const char* e = getenv("PFPP_GEMM_DBG");
    switch (e ? atoi(e) : 0) {
      case 1: return launch_pl<MT, NT, WM, WN, NS, 1>(p0, batch, st, group_m);
      case 3: return launch_pl<MT, NT, WM, WN, NS, 3>(p0, batch, st, group_m);
      case 7: return launch_pl<MT, NT, WM, WN, NS, 7>(p0, batch, st, group_m);
      case 15: return launch_pl<MT, NT, WM, WN, NS, 15>(p0, batch, st, group_m);
      case 23: return launch_pl<MT, NT, WM, WN, NS, 23>(p0, batch, st, group_m);
      case 128: return launch_pl<MT, NT, WM, WN, NS, 128>(p0, batch, st, group_m);
      default: break;
    }
  }
#endif
  static bool attr_set = false;
  auto kern = gemm_pl_kernel<MT, NT, WM, WN, NS, DBG>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmP p = p0;
  p.tiles_m = (p.M + C::BM - 1) / C::BM;
  p.tiles_n = (p.N + C::BN - 1) / C::BN;
  p.group_m = p.tiles_n > 1 ? group_m : 0;
  p.split_k = 1;
  p.k_chunk = 0;
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), C::SMEM, st, p);
  return pfpp::check_launch("pfpp_gemm");
}

}  // namespace pl

// variant: 0 = pick by shape, 1 = 256x256 (8 waves of 128x64, 2 stages), 2 = 256x128 (8 waves, 3 stages), 3 = 128x128 (4 waves,
// 2 stages, two workgroups per CU), 4 / 6 = 128x64 (2 / 3 stages), 5 = 64x128: small tiles for narrow outputs
int launch_f16x3_planes(const GemmP& p, int batch, hipStream_t st, int group_m, int variant) {
  if (variant == 0) {
    const int64_t t256 = ((int64_t)(p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
    const int64_t t21 = ((int64_t)(p.M + 255) / 256) * ((p.N + 127) / 128) * batch;
    variant = t256 >= 512 ? 1 : (t21 >= 224 ? 2 : 3);
  }
  switch (variant) {
    case 1: return pl::launch_pl<4, 2, 2, 4, 2>(p, batch, st, group_m);
    case 2: return pl::launch_pl<2, 2, 4, 2, 3>(p, batch, st, group_m);
    case 4: return pl::launch_pl<2, 1, 2, 2, 2>(p, batch, st, group_m);      // 128x64, 4 waves of 64x32
    case 5: return pl::launch_pl<1, 2, 2, 2, 2>(p, batch, st, group_m);      // 64x128, 4 waves of 32x64
    case 6: return pl::launch_pl<2, 1, 2, 2, 3>(p, batch, st, group_m);      // 128x64, 3 stages
    case 7: return pl::launch_pl<2, 1, 2, 2, 4>(p, batch, st, group_m);
    case 8: return pl::launch_pl<2, 1, 2, 2, 6>(p, batch, st, group_m);
    case 9: return pl::launch_pl<2, 2, 2, 2, 3>(p, batch, st, group_m);      // 128x128, 3 / 4 stages (one workgroup per CU)
    case 10: return pl::launch_pl<2, 2, 2, 2, 4>(p, batch, st, group_m);
    default: return pl::launch_pl<2, 2, 2, 2, 2>(p, batch, st, group_m);
  }
}

}  // namespace pfpp_gemm_detail
