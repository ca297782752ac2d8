// probe: a split-f16 GEMM main loop whose WEIGHT operand never touches LDS.  C[M, N] = A[M, K] . W[N, K]^T with the activations as
// the row-major hi / lo planes every producer of the product writes today, and the weights FRAGMENT-BLOCKED
// ([N / 32][K / 16][64 lanes][8 halfs]: one MFMA operand fragment = one contiguous 1 KB; the layout of pfpp_pw.fhi / flo).
// Workgroup = 4 waves = (32 MT) rows x (128 NT) columns of C; wave w owns the NT 32-column units 4 NT by + NT w .. and ALL rows of
// the tile: its weight fragments come straight from global memory into registers (D K-tiles deep), the activation tile goes through
// a D-stage LDS-DMA ring shared by the four waves (8 KB per stage at MT = 2: the ring can be deep).
// Per 32-deep K-tile and CU at MT = 2, NT = 1: 48 MFMAs (384 cycles / SIMD), 40 KB through LDS (the product kernel's 128 x 64 tile:
// 72 KB), 24 KB through the texture path.  Arithmetic = gemm_pl_kernel's (lo.hi, hi.lo, hi.hi per 16-deep step, k ascending).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lab/gemm_wdirect_probe.hip -o tools/lab/_bin/gemm_wdirect_probe
//   tools/lab/_bin/gemm_wdirect_probe 3850 512 512
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
#ifndef MT_
#define MT_ 2      // 32-row tiles per workgroup (every wave multiplies all of them)
#endif
#ifndef NT_
#define NT_ 1      // 32-column units per wave
#endif
#ifndef DEPTH
#define DEPTH 4    // K-tiles in the pipeline: LDS stages of the activation ring = register slots of the weight fragments; DEPTH - 1 in flight
#endif
#ifndef XCD
#define XCD 1      // 1-D grid remapped so that an XCD owns consecutive tiles (row-major: the column tiles of a row panel share an L2)
#endif
#ifndef ABL
#define ABL 0      // ablations: 1 no MFMA, 2 no weight loads after the prologue, 4 no DMA after the prologue, 8 no barrier, 32 no epilogue stores
#endif
#ifndef KT_
#define KT_ 1      // K-tile depth in units of 32: 2 = 64-deep stages (one barrier per 24 matrix instructions at MT 2, whole 128-byte lines of A per row)
#endif
#ifndef IL
#define IL 0       // PF form: 1 = the next tile's LDS reads and the new tile's loads are spread between the matrix instructions
#endif
#ifndef PF
#define PF 0       // 1 = the activation fragments of tile kt + 1 are read from LDS while tile kt is multiplied (register double buffer; DEPTH even)
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ half8 lds_rd(uint32_t addr) {
  half8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ half8 gld(const half8* p) {
  half8 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int MT, int NT, int D, int KT>
__global__ __launch_bounds__(256) void wdirect_kernel(const _Float16* __restrict__ a_hi, const _Float16* __restrict__ a_lo,
                                                      const half8* __restrict__ w_fh, const half8* __restrict__ w_fl,
                                                      float* __restrict__ C, int M, int N, int K) {
  constexpr int ROWB = 64 * KT;                                        // bytes of a row of one plane in a stage
  constexpr int BM = 32 * MT, PLANE = BM * ROWB, STAGE = 2 * PLANE;    // bytes: BM rows of 32 KT halfs, two planes
  constexpr int NPW = MT * KT;                                         // 1 KB DMA pieces (16 / KT rows of one plane) per wave and stage
  constexpr int P = NPW + 4 * NT * KT;                                 // vector-memory operations a wave issues per K-tile
  constexpr int CH = 4 * KT, RPP = 64 / CH;                            // 16-byte chunks per row, rows per piece
  constexpr int NS = 2 * KT;                                           // 16-deep steps per K-tile
  extern __shared__ __align__(1024) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int KB = K / 16, nk = K / (32 * KT);
  const int tiles_n = N / (128 * NT);
#if XCD
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q_ = nwg >> 3, r_ = nwg & 7;
  const int tile = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + local;
#else
  const int tile = blockIdx.x;
#endif
  const int bx = tile / tiles_n, by = tile - bx * tiles_n;
  const int m0 = bx * BM;
  const int nb0 = (by * 4 + wave) * NT;                                // this wave's first 32-column unit
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;

  // ---- activation tile: piece q = wave + 4 j = 16 rows of one plane; lane i -> row i >> 2, physical chunk i & 3 holding the row's
  //      logical 16-byte chunk (i & 3) ^ ((row >> 2) & 3)  (the swizzle of gemm_pl.hip: conflict-free ds_read_b128)
  const char* src[NPW];
  uint32_t dst[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int q = wave + 4 * j;                                        // piece: plane q / (PLANE / 1024), rows RPP (q % ..) ..
    const int pl = q / (PLANE / 1024), row = (q % (PLANE / 1024)) * RPP + lane / CH;
    const int sw_ = KT == 1 ? ((row >> 2) & 3) : ((row >> 1) & 7);
    const int chunk = (lane % CH) ^ sw_;
    const int grow = m0 + row < M ? m0 + row : M - 1;
    src[j] = reinterpret_cast<const char*>((pl ? a_lo : a_hi) + (size_t)grow * K + chunk * 8);
    dst[j] = lds0 + q * 1024;
  }
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int j = 0; j < NPW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[j] + (size_t)kt * ROWB), (lds_void*)(uintptr_t)(dst[j] + stage * STAGE), 16, 0, 0);
  };
  // ---- weight fragments of K-tile kt: units nb0 .. nb0 + NT - 1, steps 2 kt and 2 kt + 1, both planes
  const half8* wbh[NT];
  const half8* wbl[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    wbh[j] = w_fh + (size_t)(nb0 + j) * KB * 64 + lane;
    wbl[j] = w_fl + (size_t)(nb0 + j) * KB * 64 + lane;
  }
  half8 wh[D][NT][NS], wl[D][NT][NS];
  auto wload = [&](int kt, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const half8* ph = wbh[j] + (size_t)kt * 64 * NS;
      const half8* pl = wbl[j] + (size_t)kt * 64 * NS;
      static_for<NS>([&](auto s_c) {
        constexpr int s_ = decltype(s_c)::value;
        wh[slot][j][s_] = gld<1024 * s_>(ph);
        wl[slot][j][s_] = gld<1024 * s_>(pl);
      });
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;

  const int sw = KT == 1 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
  uint32_t a_ad[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) a_ad[s] = lds0 + l31 * ROWB + (((2 * s + lhi) ^ sw) << 4);

  auto rd_frags = [&](half8 (&fh)[MT], half8 (&fl)[MT], int stage, auto s_c) {
    constexpr int s = decltype(s_c)::value;
    const uint32_t ad = a_ad[s] + stage * STAGE;
    static_for<MT>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      fh[t] = lds_rd<32 * ROWB * t>(ad);
      fl[t] = lds_rd<PLANE + 32 * ROWB * t>(ad);
    });
  };
  auto wait_frags = [&](half8 (&fh)[MT], half8 (&fl)[MT], auto left_c) {     // left = LDS reads issued behind these that may stay in flight
    constexpr int LEFT = decltype(left_c)::value;
    if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fh[0]), "+v"(fl[0]) : "n"(LEFT));
    else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fl[0]), "+v"(fl[1]) : "n"(LEFT));
    else if constexpr (MT == 3) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fh[2]), "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]) : "n"(LEFT));
    else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fh[2]), "+v"(fh[3]), "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]) : "n"(LEFT));
  };
  auto name_w = [&](auto slot_c) {      // the preceding vmcnt wait orders the uses of this slot's registers
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j)
      static_for<NS>([&](auto s_c) {
        half8 &r0 = wh[slot][j][decltype(s_c)::value], &r2 = wl[slot][j][decltype(s_c)::value];
        asm volatile("" : "+v"(r0), "+v"(r2));
      });
  };
  auto mma = [&](const half8 (&fh)[MT], const half8 (&fl)[MT], auto slot_c, auto s_c) {
    constexpr int slot = decltype(slot_c)::value, s = decltype(s_c)::value;
    if (ABL & 1) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j][0] += (float)fh[t][0] + (float)fl[t][0] + (float)wh[slot][j][s][0] + (float)wl[slot][j][s][0];
      return;
    }
    // term-major like gemm_pl_kernel: lo.hi of every tile, then hi.lo, then hi.hi (per accumulator: the same three products in the same order)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[t], wh[slot][j][s], acc[t][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], wl[slot][j][s], acc[t][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], wh[slot][j][s], acc[t][j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // K-tile kt (stage = slot = U = kt % D): wait for its own loads (the tile after it may be in flight behind them), barrier (every
  // wave's DMA pieces of tile kt have landed AND every wave is through tile kt - 1, whose stage the next DMA overwrites), request
  // tile kt + D - 1, multiply.
  half8 ffh[NS][MT], ffl[NS][MT];
  auto ktile = [&](int kt, auto u_c) {
    constexpr int U = decltype(u_c)::value;
    constexpr int UN = (U + D - 1) % D;
    // outstanding behind tile kt's loads: the loads of tiles kt + 1 .. kt + D - 2 (those that exist)
    const int behind = min(D - 2, nk - 1 - kt);
    if (!(ABL & 8)) {
      if (D >= 3 && behind >= D - 2) wait_vmcnt<(D - 2) * P>();
      else if (D >= 4 && behind == D - 3) wait_vmcnt<(D >= 4 ? D - 3 : 0) * P>();
      else if (D >= 5 && behind == D - 4) wait_vmcnt<(D >= 5 ? D - 4 : 0) * P>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    name_w(u_c);
    if (kt + D - 1 < nk) {
      if (!(ABL & 4)) dma(kt + D - 1, UN);
      if (!(ABL & 2)) wload(kt + D - 1, std::integral_constant<int, UN>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NS>([&](auto s_c) { rd_frags(ffh[decltype(s_c)::value], ffl[decltype(s_c)::value], U, s_c); });
    static_for<NS>([&](auto s_c) {
      constexpr int s_ = decltype(s_c)::value;
      wait_frags(ffh[s_], ffl[s_], std::integral_constant<int, 2 * MT * (NS - 1 - s_)>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(ffh[s_], ffl[s_], u_c, s_c);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  if constexpr (PF == 0) {
  // prologue: tiles 0 .. D - 2 requested
  static_for<D - 1>([&](auto u_c) {
    constexpr int U = decltype(u_c)::value;
    if (U < nk) { dma(U, U); wload(U, u_c); }
  });
  int kt = 0;
  for (; kt + D <= nk; kt += D) static_for<D>([&](auto u_c) { ktile(kt + decltype(u_c)::value, u_c); });
  static_for<D - 1>([&](auto u_c) { if (kt + decltype(u_c)::value < nk) ktile(kt + decltype(u_c)::value, u_c); });
  } else {
  // ---- software-pipelined form: the fragments of tile kt + 1 are requested from LDS in front of tile kt's matrix instructions
  static_assert(PF == 0 || (D % 2 == 0 && KT == 1), "register double buffer: even depth, 32-deep tiles");
  half8 gh[2][2][MT], gl[2][2][MT];                 // [parity of the tile][16-deep step][row block]
  auto wait_all = [&](auto par_c) {
    constexpr int par = decltype(par_c)::value;
    static_for<2>([&](auto s_c) { wait_frags(gh[par][decltype(s_c)::value], gl[par][decltype(s_c)::value], I0{}); });
  };
  // IL == 2 (MT 2, NT 1): one memory instruction in the shadow of every matrix instruction — a wave stalled in the issue of a vector-memory
  // instruction (texture queue full) issues no matrix instruction either; used for the tiles that still request a tile and read a next one
  auto ktile_il2 = [&](int kt, auto u_c) {
    constexpr int U = decltype(u_c)::value;
    constexpr int UN = (U + D - 1) % D, U1 = (U + 1) % D;
    constexpr int par = U & 1, npar = par ^ 1;
    wait_vmcnt<(D >= 4 ? D - 3 : 0) * P>();
    __builtin_amdgcn_s_barrier();
    name_w(u_c);
    const uint32_t ad0 = a_ad[0] + U1 * STAGE, ad1 = a_ad[1] + U1 * STAGE;
    const half8* ph = wbh[0] + (size_t)(kt + D - 1) * 64 * NS;
    const half8* pl = wbl[0] + (size_t)(kt + D - 1) * 64 * NS;
    auto mem = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i == 0) gh[npar][0][0] = lds_rd<0>(ad0);
      else if constexpr (i == 1) gl[npar][0][0] = lds_rd<PLANE>(ad0);
      else if constexpr (i == 2) gh[npar][0][1] = lds_rd<32 * ROWB>(ad0);
      else if constexpr (i == 3) gl[npar][0][1] = lds_rd<PLANE + 32 * ROWB>(ad0);
      else if constexpr (i == 4) gh[npar][1][0] = lds_rd<0>(ad1);
      else if constexpr (i == 5) gl[npar][1][0] = lds_rd<PLANE>(ad1);
      else if constexpr (i == 6) gh[npar][1][1] = lds_rd<32 * ROWB>(ad1);
      else if constexpr (i == 7) gl[npar][1][1] = lds_rd<PLANE + 32 * ROWB>(ad1);
      else if constexpr (i == 8) __builtin_amdgcn_global_load_lds((gbl_void*)(src[0] + (size_t)(kt + D - 1) * ROWB), (lds_void*)(uintptr_t)(dst[0] + UN * STAGE), 16, 0, 0);
      else if constexpr (i == 9) __builtin_amdgcn_global_load_lds((gbl_void*)(src[1] + (size_t)(kt + D - 1) * ROWB), (lds_void*)(uintptr_t)(dst[1] + UN * STAGE), 16, 0, 0);
      else if constexpr (i == 10) wh[UN][0][0] = gld<0>(ph);
      else if constexpr (i == 11) wh[UN][0][1] = gld<1024>(ph);
      else if constexpr (i == 12) wl[UN][0][0] = gld<0>(pl);
      else if constexpr (i == 13) wl[UN][0][1] = gld<1024>(pl);
    };
    static_for<12>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value, s_ = i / 6, r = i % 6, term = r / 2, t = r % 2;
      __builtin_amdgcn_sched_barrier(0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? gl[par][s_][t] : gh[par][s_][t], term == 1 ? wl[U][0][s_] : wh[U][0][s_], acc[t][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mem(i_c);
    });
    __builtin_amdgcn_sched_barrier(0);
    mem(std::integral_constant<int, 12>{});
    mem(std::integral_constant<int, 13>{});
    __builtin_amdgcn_sched_barrier(0);
    wait_all(std::integral_constant<int, npar>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto ktile_pf = [&](int kt, auto u_c) {
    constexpr int U = decltype(u_c)::value;
    constexpr int UN = (U + D - 1) % D, U1 = (U + 1) % D;
    constexpr int par = U & 1, npar = par ^ 1;
    // tile kt + 1 has to have landed: the loads of tiles kt + 2 .. kt + D - 2 (those that exist) may stay in flight
    const int behind = max(0, min(D - 3, nk - 2 - kt));
    if (!(ABL & 8)) {
      if (D >= 4 && behind >= D - 3) wait_vmcnt<(D >= 4 ? D - 3 : 0) * P>();
      else if (D >= 6 && behind == D - 4) wait_vmcnt<(D >= 6 ? D - 4 : 0) * P>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    name_w(u_c);
    if (IL == 1) {
      // 6 matrix instructions (s = 0) right behind the barrier, the memory instructions in their shadow, then the other 6
      __builtin_amdgcn_sched_barrier(0);
      mma(gh[par][0], gl[par][0], u_c, I0{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + D - 1 < nk) {
      if (!(ABL & 4)) dma(kt + D - 1, UN);
      if (!(ABL & 2)) wload(kt + D - 1, std::integral_constant<int, UN>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 64) && kt + 1 < nk) {
      rd_frags(gh[npar][0], gl[npar][0], U1, I0{});
      rd_frags(gh[npar][1], gl[npar][1], U1, I1{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (IL != 1) mma(gh[par][0], gl[par][0], u_c, I0{});
    mma(gh[par][1], gl[par][1], u_c, I1{});
    __builtin_amdgcn_sched_barrier(0);
    wait_all(std::integral_constant<int, npar>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<D - 1>([&](auto u_c) {
    constexpr int U = decltype(u_c)::value;
    if (U < nk) { dma(U, U); wload(U, u_c); }
  });
  {
    const int behind = min(D - 2, nk - 1);
    if (behind >= D - 2) wait_vmcnt<(D - 2) * P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    rd_frags(gh[0][0], gl[0][0], 0, I0{});
    rd_frags(gh[0][1], gl[0][1], 0, I1{});
    if (ABL & 64) { rd_frags(gh[1][0], gl[1][0], 0, I0{}); rd_frags(gh[1][1], gl[1][1], 0, I1{}); wait_all(I1{}); }
    wait_all(I0{});
  }
  int kt = 0;
  if constexpr (IL == 2 && MT == 2 && NT == 1 && KT == 1 && ABL == 0) {
    for (; kt + 2 * D - 1 <= nk; kt += D) static_for<D>([&](auto u_c) { ktile_il2(kt + decltype(u_c)::value, u_c); });
    for (; kt < nk; ++kt) static_for<D>([&](auto u_c) { if (kt % D == decltype(u_c)::value) ktile_pf(kt, u_c); });
  } else {
  for (; kt + D <= nk; kt += D) static_for<D>([&](auto u_c) { ktile_pf(kt + decltype(u_c)::value, u_c); });
  static_for<D - 1>([&](auto u_c) { if (kt + decltype(u_c)::value < nk) ktile_pf(kt + decltype(u_c)::value, u_c); });
  }
  }

  // plain epilogue (probe): lane = column, register e = row (e & 3) + 8 (e >> 2) + 4 lhi
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = (nb0 + j) * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if ((ABL & 32) ? (acc[t][j][e] == 12345.678f) : (row < M && col < N)) C[(size_t)row * N + col] = acc[t][j][e];
      }
    }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3850, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  constexpr int MT = MT_, NT = NT_, D = DEPTH, KT = KT_;
  if (N % (128 * NT) || K % (32 * KT)) { printf("N %% %d or K %% 32\n", 128 * NT); return 1; }
  const int NB = N / 32, KB = K / 16;
  std::vector<float> A((size_t)M * K), W((size_t)N * K);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : W) v = rnd() / sqrtf((float)K);
  std::vector<_Float16> ahi(A.size()), alo(A.size()), whi((size_t)NB * KB * 512), wlo(whi.size());
  for (size_t i = 0; i < A.size(); ++i) { const _Float16 h = (_Float16)A[i]; ahi[i] = h; alo[i] = (_Float16)(A[i] - (float)h); }
  for (int rb = 0; rb < NB; ++rb) for (int kb = 0; kb < KB; ++kb) for (int ln = 0; ln < 64; ++ln) for (int q = 0; q < 8; ++q) {
    const float x = W[(size_t)(rb * 32 + (ln & 31)) * K + kb * 16 + (ln >> 5) * 8 + q];
    const _Float16 h = (_Float16)x;
    const size_t o = (((size_t)rb * KB + kb) * 64 + ln) * 8 + q;
    whi[o] = h; wlo[o] = (_Float16)(x - (float)h);
  }
  _Float16 *d_ah, *d_al, *d_wh, *d_wl; float* d_c;
  CK(hipMalloc(&d_ah, ahi.size() * 2)); CK(hipMalloc(&d_al, alo.size() * 2)); CK(hipMalloc(&d_wh, whi.size() * 2)); CK(hipMalloc(&d_wl, wlo.size() * 2));
  CK(hipMalloc(&d_c, (size_t)M * N * 4));
  CK(hipMemcpy(d_ah, ahi.data(), ahi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_al, alo.data(), alo.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wh, whi.data(), whi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wl, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice));
  const int tiles = ((M + 32 * MT - 1) / (32 * MT)) * (N / (128 * NT));
  const size_t smem = (size_t)D * 2 * 32 * MT * 64 * KT;
  auto kern = wdirect_kernel<MT, NT, D, KT>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, 0, d_ah, d_al, (const half8*)d_wh, (const half8*)d_wl, d_c, M, N, K); };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e0));
  const int it = 50;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> Cc((size_t)M * N);
  CK(hipMemcpy(Cc.data(), d_c, Cc.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int t = 0; t < 256; ++t) {
    const int m = t < 8 ? M - 1 - t : (int)((unsigned)(t * 2654435761u) % M), n = (int)((unsigned)(t * 40503u + 17) % N);
    double ref = 0.0;
    for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    worst = fmax(worst, fabs(ref - Cc[(size_t)m * N + n])); scale = fmax(scale, fabs(ref));
  }
  unsigned long long hsh = 1469598103934665603ull;
  for (size_t i = 0; i < Cc.size(); ++i) { unsigned u; memcpy(&u, &Cc[i], 4); hsh = (hsh ^ u) * 1099511628211ull; }
  const double us = ms / it * 1e3;
  printf("MT %d NT %d D %d KT %d XCD %d ABL %d PF %d IL %d | M %d N %d K %d: %.1f us per launch, %.1f TFLOP/s (fp32-grade), %d workgroups, LDS %zu B, max |err| %.2e of %.2e, hash %016llx\n",
         MT, NT, D, KT, XCD, ABL, PF, IL, M, N, K, us, 2.0 * M * N * K / us * 1e-6, tiles, smem, worst, scale, hsh);
  return 0;
}
