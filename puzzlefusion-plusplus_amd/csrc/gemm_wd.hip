// Plane GEMM whose WEIGHT operand never touches LDS: out = (A . W^T) / scale + bias + residual.
//
// Reference: the linear layers of EncoderLayer.forward (denoiser/model/modules/attention.py:77-90: attn.to_q|k|v, attn.to_out[0],
// ff.net[2], the latter two followed by the residual add), eval mode, as sequenced by pfpp_tlayers_eval above the few-token range.
//
// Why a kernel of its own.  The tiled plane GEMM (gemm_pl.hip) stages BOTH operands through its LDS-DMA ring: a 128 x 64 tile moves
// 72 KB through LDS per 32-deep K-tile for 384 cycles of matrix work per SIMD — the loop is LDS-bound, and the ring (24 KB per stage)
// cannot be deep.  Eval weights are static, so their FRAGMENT-BLOCKED planes exist (include/pfpp.h pfpp_pw.fhi / flo: one 1 KB block =
// the 64 lanes' B operands of one v_mfma_f32_32x32x16_f16).  Here
//   * a workgroup = 4 waves = (32 MT) rows x (128 NT) columns; wave w owns NT 32-column units and ALL rows of the tile: its weight
//     fragments come straight from global memory into registers (one fully coalesced 1 KB load each, D K-tiles deep, no LDS pass, no
//     sharing between the waves of a workgroup), the activation tile (row-major hi / lo planes, as every producer writes them) goes
//     through a D-stage LDS-DMA ring shared by the four waves — 8 KB per stage at MT = 2;
//   * per K-tile at MT = 2, NT = 1: 40 KB through LDS (72), 24 KB through the texture path, the same 48 matrix instructions;
//   * one barrier per K-tile: it publishes the DMA pieces of tile kt and frees the stage tile kt + D - 1 lands in;
//   * epilogue through a wave-private LDS patch (ds_write_b32 in accumulator order, ds_read_b128 along the rows): 16-byte stores.
// Measured stand-alone (tools/lab/gemm_wdirect_probe.hip, 3850 rows): 512 x 512 10.1 us, 512 x 2048 29.6 us, 1536 x 512 27.1 us against
// 19.2 / 48.0 / 33.0 us of the tiled kernel inside the step.
// Arithmetic: the three split-f16 products of a 16-deep step in the order lo.hi, hi.lo, hi.hi, k ascending, one accumulator chain per
// output, epilogue (acc * alpha + bias) + residual — gemm_pl_kernel's, bit for bit (tested).
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "pfpp_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct WdP {
  const _Float16 *ah, *al; int64_t lda;      // planes of a_scale * A [M, K]
  const half8 *fh, *fl;                      // fragment-blocked planes of w_scale * W [N, K]
  float alpha;                               // 1 / (a_scale * w_scale)
  const float* bias;                         // [N] or null
  const float* res; int64_t ldr;             // [M, ldr] or null (may alias out)
  float* out; int64_t ldc;
  int M, N, K;
};

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

template <int MT, int NT, int D, bool X1 = false>
struct WdCfg {
  static constexpr int NPL = X1 ? 1 : 2;                       // planes per operand: the single-pass fp16 mode reads the hi planes only
  static constexpr int BM = 32 * MT, BN = 128 * NT;
  static constexpr int PLANE = BM * 64, STAGE = NPL * PLANE;   // bytes: BM rows of 32 halfs per plane
  static constexpr int PATCH = 4096;                           // per wave: 32 rows x 32 floats
  static constexpr size_t SMEM = (size_t)D * STAGE + 4 * PATCH;
};

template <int MT, int NT, int D, bool X1, bool PF>
__device__ __forceinline__ void gemm_wd_body(const WdP& p) {
  pfpp_chain_prio();
  using C = WdCfg<MT, NT, D, X1>;
  constexpr int BM = C::BM, PLANE = C::PLANE, STAGE = C::STAGE, NPL = C::NPL;
  constexpr int NPW = NPL * MT / 2;                            // 1 KB DMA pieces (16 rows of one plane) per wave and stage
  constexpr int P = NPW + 2 * NPL * NT;                        // vector-memory operations a wave issues per K-tile
  static_assert(NPW >= 1, "at least one DMA piece per wave");
  extern __shared__ __align__(1024) char wd_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int KB = p.K / 16, nk = p.K / 32;
  const int tiles_n = p.N / C::BN;
  // 1-D grid; the hardware deals workgroup ids to the 8 XCDs round-robin: remapped so that an XCD owns consecutive tiles (row-major:
  // the column tiles of a row panel share one L2; the weights are read by every row panel anyway)
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q_ = nwg >> 3, r_ = nwg & 7;
  const int tile = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + local;
  const int bx = tile / tiles_n, by = tile - bx * tiles_n;
  const int m0 = bx * BM;
  const int nb0 = (by * 4 + wave) * NT;                        // this wave's first 32-column unit
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)wd_smem;

  // ---- activation tile: piece q = wave + 4 j = 16 rows of one plane; lane i -> row i >> 2, physical chunk i & 3 holding the row's
  //      logical 16-byte chunk (i & 3) ^ ((row >> 2) & 3)  (gemm_pl.hip's swizzle: conflict-free ds_read_b128); rows past M repeat row M - 1
  const char* src[NPW];
  uint32_t dst[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int q = wave + 4 * j;                                // (single-pass: pieces 0 .. 2 MT - 1, the hi plane only)
    const int pl = q / (2 * MT), row = (q % (2 * MT)) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t grow = m0 + row < p.M ? m0 + row : p.M - 1;
    src[j] = reinterpret_cast<const char*>((pl ? p.al : p.ah) + grow * p.lda + chunk * 8);
    dst[j] = lds0 + q * 1024;
  }
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int j = 0; j < NPW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[j] + (size_t)kt * 64), (lds_void*)(uintptr_t)(dst[j] + stage * STAGE), 16, 0, 0);
  };
  // ---- weight fragments of K-tile kt: units nb0 .. nb0 + NT - 1, 16-deep steps 2 kt and 2 kt + 1, both planes
  const half8* wbh[NT];
  const half8* wbl[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    wbh[j] = p.fh + (size_t)(nb0 + j) * KB * 64 + lane;
    wbl[j] = p.fl + (size_t)(nb0 + j) * KB * 64 + lane;
  }
  half8 wh[D][NT][2], wl[D][NT][2];
  auto wload = [&](int kt, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const half8* ph = wbh[j] + (size_t)kt * 128;
      const half8* pl = wbl[j] + (size_t)kt * 128;
      wh[slot][j][0] = gld<0>(ph);
      wh[slot][j][1] = gld<1024>(ph);
      if constexpr (!X1) {
        wl[slot][j][0] = gld<0>(pl);
        wl[slot][j][1] = gld<1024>(pl);
      }
    }
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;

  const int sw = (l31 >> 2) & 3;
  uint32_t a_ad[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) a_ad[s] = lds0 + l31 * 64 + (((2 * s + lhi) ^ sw) << 4);

  auto rd_frags = [&](half8 (&fh)[MT], half8 (&fl)[MT], int stage, auto s_c) {
    const uint32_t ad = a_ad[decltype(s_c)::value] + stage * STAGE;
    static_for<MT>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      fh[t] = lds_rd<2048 * t>(ad);
      if constexpr (!X1) fl[t] = lds_rd<PLANE + 2048 * t>(ad);
    });
  };
  auto wait_frags = [&](half8 (&fh)[MT], half8 (&fl)[MT], auto left_c) {     // left = LDS reads issued behind these that may stay in flight
    constexpr int LEFT = decltype(left_c)::value;
    static_assert(MT == 2 || MT == 4, "row tiles per workgroup");
    if constexpr (X1 && MT == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fh[0]), "+v"(fh[1]) : "n"(LEFT));
    else if constexpr (X1) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fh[2]), "+v"(fh[3]) : "n"(LEFT));
    else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fl[0]), "+v"(fl[1]) : "n"(LEFT));
    else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fh[0]), "+v"(fh[1]), "+v"(fh[2]), "+v"(fh[3]), "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]) : "n"(LEFT));
  };
  auto name_w = [&](auto slot_c) {      // the vmcnt wait in front orders the uses of this slot's registers: name them behind it
    constexpr int slot = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      half8 &r0 = wh[slot][j][0], &r1 = wh[slot][j][1], &r2 = wl[slot][j][0], &r3 = wl[slot][j][1];
      if constexpr (X1) asm volatile("" : "+v"(r0), "+v"(r1));
      else asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    }
  };
  // term-major like gemm_pl_kernel: lo.hi of every tile, then hi.lo, then hi.hi (per accumulator: the same products in the same order)
  auto mma = [&](const half8 (&fh)[MT], const half8 (&fl)[MT], auto slot_c, auto s_c) {
    constexpr int slot = decltype(slot_c)::value, s = decltype(s_c)::value;
    if constexpr (!X1) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[t], wh[slot][j][s], acc[t][j], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], wl[slot][j][s], acc[t][j], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], wh[slot][j][s], acc[t][j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // K-tile kt (stage = slot = U = kt % D): wait for its own loads (the D - 2 tiles behind it stay in flight), barrier (every wave's DMA
  // pieces of tile kt have landed AND every wave is through tile kt - 1, whose stage / slot the next request overwrites), request tile
  // kt + D - 1, multiply.
  half8 f0h[MT], f0l[MT], f1h[MT], f1l[MT];
  auto ktile = [&](int kt, auto u_c) {
    constexpr int U = decltype(u_c)::value;
    constexpr int UN = (U + D - 1) % D;
    const int behind = min(D - 2, nk - 1 - kt);
    if (behind >= D - 2) wait_vmcnt<(D - 2) * P>();
    else if (D >= 4 && behind == D - 3) wait_vmcnt<(D >= 4 ? D - 3 : 0) * P>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    name_w(u_c);
    if (kt + D - 1 < nk) {
      dma(kt + D - 1, UN);
      wload(kt + D - 1, std::integral_constant<int, UN>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    rd_frags(f0h, f0l, U, I0{});
    rd_frags(f1h, f1l, U, I1{});
    wait_frags(f0h, f0l, std::integral_constant<int, NPL * MT>{});
    __builtin_amdgcn_sched_barrier(0);
    mma(f0h, f0l, u_c, I0{});
    __builtin_amdgcn_sched_barrier(0);
    wait_frags(f1h, f1l, I0{});
    mma(f1h, f1l, u_c, I1{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_for<D - 1>([&](auto u_c) {      // prologue: tiles 0 .. D - 2 requested
    constexpr int U = decltype(u_c)::value;
    if (U < nk) { dma(U, U); wload(U, u_c); }
  });
  if constexpr (!PF) {
    int kt = 0;
    for (; kt + D <= nk; kt += D) static_for<D>([&](auto u_c) { ktile(kt + decltype(u_c)::value, u_c); });
    static_for<D - 1>([&](auto u_c) { if (kt + decltype(u_c)::value < nk) ktile(kt + decltype(u_c)::value, u_c); });
  } else {
    // ---- software-pipelined form (round 5; 64 x 128 tile, three workgroups' worth of tiles per CU: the 1536- / 2048-wide outputs).
    // A wave that stalls in the issue of a vector-memory instruction (texture queue full) issues no matrix instruction either, and the
    // loop above issues its six loads and eight LDS reads in one block in front of the twelve matrix instructions.  Here the fragments of
    // tile kt + 1 are read from LDS while tile kt is multiplied (register double buffer), and every memory instruction sits in the shadow
    // of one matrix instruction: 26.4 -> 23.5 us at 3850 x 1536 x 512, 34.4 -> 30.2 at 16000 x 512 x 512; no gain where one workgroup has
    // a CU to itself (512-wide outputs at 3,850 rows: those keep the loop above).  Same products in the same order per accumulator.
    static_assert(!PF || (MT == 2 && NT == 1 && !X1 && D == 4), "pipelined form: 64 x 128 tile, split-f16, four stages");
    half8 gh[2][2][MT], gl[2][2][MT];                 // [parity of the tile][16-deep step][row block]
    auto wait_all = [&](auto par_c) {
      constexpr int par = decltype(par_c)::value;
      wait_frags(gh[par][0], gl[par][0], I0{});
      wait_frags(gh[par][1], gl[par][1], I0{});
    };
    auto ktile_il = [&](int kt, auto u_c) {           // tiles that still request tile kt + D - 1 and read tile kt + 1
      constexpr int U = decltype(u_c)::value;
      constexpr int UN = (U + D - 1) % D, U1 = (U + 1) % D;
      constexpr int par = U & 1, npar = par ^ 1;
      wait_vmcnt<(D - 3) * P>();                      // tile kt + 1 has landed (tile kt + 2 may be in flight)
      __builtin_amdgcn_s_barrier();
      name_w(u_c);
      const uint32_t ad0 = a_ad[0] + U1 * STAGE, ad1 = a_ad[1] + U1 * STAGE;
      const half8* ph = wbh[0] + (size_t)(kt + D - 1) * 128;
      const half8* pl = wbl[0] + (size_t)(kt + D - 1) * 128;
      auto mem = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        if constexpr (i == 0) gh[npar][0][0] = lds_rd<0>(ad0);
        else if constexpr (i == 1) gl[npar][0][0] = lds_rd<PLANE>(ad0);
        else if constexpr (i == 2) gh[npar][0][1] = lds_rd<2048>(ad0);
        else if constexpr (i == 3) gl[npar][0][1] = lds_rd<PLANE + 2048>(ad0);
        else if constexpr (i == 4) gh[npar][1][0] = lds_rd<0>(ad1);
        else if constexpr (i == 5) gl[npar][1][0] = lds_rd<PLANE>(ad1);
        else if constexpr (i == 6) gh[npar][1][1] = lds_rd<2048>(ad1);
        else if constexpr (i == 7) gl[npar][1][1] = lds_rd<PLANE + 2048>(ad1);
        else if constexpr (i == 8) __builtin_amdgcn_global_load_lds((gbl_void*)(src[0] + (size_t)(kt + D - 1) * 64), (lds_void*)(uintptr_t)(dst[0] + UN * STAGE), 16, 0, 0);
        else if constexpr (i == 9) __builtin_amdgcn_global_load_lds((gbl_void*)(src[1] + (size_t)(kt + D - 1) * 64), (lds_void*)(uintptr_t)(dst[1] + UN * STAGE), 16, 0, 0);
        else if constexpr (i == 10) wh[UN][0][0] = gld<0>(ph);
        else if constexpr (i == 11) wh[UN][0][1] = gld<1024>(ph);
        else if constexpr (i == 12) wl[UN][0][0] = gld<0>(pl);
        else if constexpr (i == 13) wl[UN][0][1] = gld<1024>(pl);
      };
      static_for<12>([&](auto i_c) {                  // term-major per 16-deep step, as mma(): lo.hi, hi.lo, hi.hi of both row blocks
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
    auto ktile_tail = [&](int kt, auto u_c) {         // the last tiles: requests and reads only where a tile is left
      constexpr int U = decltype(u_c)::value;
      constexpr int UN = (U + D - 1) % D, U1 = (U + 1) % D;
      constexpr int par = U & 1, npar = par ^ 1;
      if (nk - 2 - kt >= D - 3) wait_vmcnt<(D - 3) * P>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      name_w(u_c);
      if (kt + D - 1 < nk) {
        dma(kt + D - 1, UN);
        wload(kt + D - 1, std::integral_constant<int, UN>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        rd_frags(gh[npar][0], gl[npar][0], U1, I0{});
        rd_frags(gh[npar][1], gl[npar][1], U1, I1{});
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(gh[par][0], gl[par][0], u_c, I0{});
      mma(gh[par][1], gl[par][1], u_c, I1{});
      __builtin_amdgcn_sched_barrier(0);
      wait_all(std::integral_constant<int, npar>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nk - 1 >= D - 2) wait_vmcnt<(D - 2) * P>();   // tile 0 has landed
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    rd_frags(gh[0][0], gl[0][0], 0, I0{});
    rd_frags(gh[0][1], gl[0][1], 0, I1{});
    wait_all(I0{});
    int kt = 0;
    for (; kt + 2 * D - 1 <= nk; kt += D) static_for<D>([&](auto u_c) { ktile_il(kt + decltype(u_c)::value, u_c); });
    for (; kt < nk; ++kt) static_for<D>([&](auto u_c) { if (kt % D == decltype(u_c)::value) ktile_tail(kt, u_c); });
  }

  // ---- epilogue: (acc * alpha + bias) + residual, every tile through the wave's private 4 KB patch (behind the ring: no barrier needed)
  const uint32_t patch = lds0 + D * STAGE + wave * C::PATCH;
  const float alpha = p.alpha;
  const int rcol = (lane & 7) * 4, rrow = lane >> 3;           // read side: lane -> 4 floats at column 4 (lane % 8) of rows lane / 8 + 8 k
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col_w = (nb0 + j) * 32;
    const float sh = p.bias ? p.bias[col_w + l31] : 0.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[t][j][e] * alpha;
        const float w = v + sh;
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lhi;
        asm volatile("ds_write_b32 %0, %1" ::"v"(patch + r * 128 + l31 * 4), "v"(w) : "memory");
      }
      f32x4 vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(vv[k]) : "v"(patch + (rrow + 8 * k) * 128 + rcol * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = m0 + t * 32 + rrow + 8 * k;
        if (row < p.M) {
          float4 v = make_float4(vv[k][0], vv[k][1], vv[k][2], vv[k][3]);
          if (p.res) {
            const float4 q = *reinterpret_cast<const float4*>(p.res + (int64_t)row * p.ldr + col_w + rcol);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          *reinterpret_cast<float4*>(p.out + (int64_t)row * p.ldc + col_w + rcol) = v;
        }
      }
    }
  }
}

template <int MT, int NT, int D, bool X1>
__global__ __launch_bounds__(256) void gemm_wd_kernel(const WdP p) { gemm_wd_body<MT, NT, D, X1, false>(p); }
// the software-pipelined loop (a kernel name of its own: the profiles' instantiation names of the plain loop stay what they were)
template <int MT, int NT, int D>
__global__ __launch_bounds__(256) void gemm_wd_pf_kernel(const WdP p) { gemm_wd_body<MT, NT, D, false, true>(p); }

// ---- fragment-blocking of row-major planes (training: the weights change every step, their blocked copies are made where they are read)
struct RbJob { const _Float16 *hi, *lo; _Float16 *fhi, *flo; int N, K; int64_t ldw; int transposed; int first_wg; };
struct RbP { RbJob job[PFPP_REBLOCK_MAX]; int n; };

// plain: piece ((nb * KB + kb) * 64 + lane) of the blocked plane = W[32 nb + lane % 32][16 kb + 8 (lane / 32) .. + 8); one thread per piece
// transposed (the blocked planes of W^T [K, N], the operand of dX = dY . W): a workgroup stages 64 rows x 64 columns of both planes in
//   LDS and writes the 2 x 4 blocks they make: piece lane of block (kb32, s) = W[16 s + 8 (lane / 32) + 0..7][32 kb32 + lane % 32]
__global__ __launch_bounds__(256) void reblock_kernel(const RbP p) {
  __shared__ __align__(16) _Float16 tile[2][64][72];
  int j = 0;
#pragma unroll
  for (int k = 1; k < PFPP_REBLOCK_MAX; ++k)
    if (k < p.n && (int)blockIdx.x >= p.job[k].first_wg) j = k;
  const RbJob& q = p.job[j];
  const int wg = blockIdx.x - q.first_wg, tid = threadIdx.x;
  if (!q.transposed) {
    const int KB = q.K / 16;
    const int64_t piece = (int64_t)wg * 256 + tid;
    if (piece >= (int64_t)(q.N / 32) * KB * 64) return;
    const int lane = (int)(piece & 63), l31 = lane & 31, lhi = lane >> 5;
    const int64_t blk = piece >> 6;
    const int nb = (int)(blk / KB), kb = (int)(blk - (int64_t)nb * KB);
    const int64_t src = (int64_t)(32 * nb + l31) * q.ldw + 16 * kb + 8 * lhi;
    reinterpret_cast<half8*>(q.fhi)[piece] = *reinterpret_cast<const half8*>(q.hi + src);
    reinterpret_cast<half8*>(q.flo)[piece] = *reinterpret_cast<const half8*>(q.lo + src);
    return;
  }
  const int tiles_k = q.K / 64;
  const int tn = wg / tiles_k, tk = wg - tn * tiles_k;      // rows 64 tn .. (contraction of W^T), columns 64 tk ..
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 3) + 32 * i, chunk = tid & 7;
    const int64_t src = (int64_t)(64 * tn + row) * q.ldw + 64 * tk + 8 * chunk;
    *reinterpret_cast<half8*>(&tile[0][row][8 * chunk]) = *reinterpret_cast<const half8*>(q.hi + src);
    *reinterpret_cast<half8*>(&tile[1][row][8 * chunk]) = *reinterpret_cast<const half8*>(q.lo + src);
  }
  __syncthreads();
  const int NB16 = q.N / 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pid = tid + 256 * i;
    const int kbl = pid >> 8, s = (pid >> 6) & 3, lane = pid & 63, l31 = lane & 31, lhi = lane >> 5;
    half8 vh, vl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      vh[e] = tile[0][16 * s + 8 * lhi + e][32 * kbl + l31];
      vl[e] = tile[1][16 * s + 8 * lhi + e][32 * kbl + l31];
    }
    const int64_t piece = ((int64_t)(2 * tk + kbl) * NB16 + (4 * tn + s)) * 64 + lane;
    reinterpret_cast<half8*>(q.fhi)[piece] = vh;
    reinterpret_cast<half8*>(q.flo)[piece] = vl;
  }
}

template <int MT, int NT, int D, bool X1 = false, bool PF = false>
int launch_wd(const WdP& p, hipStream_t st) {
  using C = WdCfg<MT, NT, D, X1>;
  static bool attr_set = false;
  if (!attr_set) {
    const void* fn;
    if constexpr (PF) fn = (const void*)gemm_wd_pf_kernel<MT, NT, D>;
    else fn = (const void*)gemm_wd_kernel<MT, NT, D, X1>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM) != hipSuccess)
      return pfpp::check_launch("pfpp_gemm_wd");
    attr_set = true;
  }
  const unsigned tiles = (unsigned)(((p.M + C::BM - 1) / C::BM) * (p.N / C::BN));
  if constexpr (PF) hipLaunchKernelGGL((gemm_wd_pf_kernel<MT, NT, D>), dim3(tiles), dim3(256), C::SMEM, st, p);
  else hipLaunchKernelGGL((gemm_wd_kernel<MT, NT, D, X1>), dim3(tiles), dim3(256), C::SMEM, st, p);
  return pfpp::check_launch("pfpp_gemm_wd");
}

}  // namespace

extern "C" int pfpp_gemm_wd_supported(int64_t M, int64_t N, int64_t K) { return M >= 1 && M <= 0x7fffffff && N >= 128 && N % 128 == 0 && K >= 32 && K % 32 == 0; }

static int gemm_wd_impl(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr,
                        float* out, int64_t ldc, int64_t M, int64_t N, int64_t K, bool single_pass, pfpp_stream_t stream) {
  PFPP_REQUIRE(A && A->hi && A->lo && w && out, "null pointer");
  PFPP_REQUIRE(w->fhi && w->flo && pfpp::aligned16(w->fhi) && pfpp::aligned16(w->flo), "the weight's fragment-blocked planes (pfpp_pw.fhi / flo) are required");
  PFPP_SUPPORTED(pfpp_gemm_wd_supported(M, N, K), "N % 128 != 0 or K % 32 != 0");
  PFPP_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 4 == 0 && (!residual || (ldr >= N && ldr % 4 == 0)), "sizes / leading dimensions");
  PFPP_REQUIRE(pfpp::aligned16(A->hi) && pfpp::aligned16(A->lo) && pfpp::aligned16(out) && pfpp::aligned16(residual) && (!bias || pfpp::aligned16(bias)),
               "16-byte aligned operands");
  WdP p;
  p.ah = (const _Float16*)A->hi; p.al = (const _Float16*)A->lo; p.lda = lda;
  p.fh = (const half8*)w->fhi; p.fl = (const half8*)w->flo;
  p.alpha = 1.0f / (A->scale * w->scale);
  p.bias = bias; p.res = residual; p.ldr = ldr; p.out = out; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  hipStream_t st = pfpp::as_stream(stream);
  // tile choice (measured, tools/lab/gemm_wdirect_probe.hip): the 128 x 256 tile once it fills the chip, the 64 x 128 tile below that
  const int64_t big_tiles = ((M + 127) / 128) * (N / 256);
  if (single_pass) {     // hi . hi only (PFPP_GEMM_F16: configs[4]'s perf mode, never for parity)
    if (N % 256 == 0 && big_tiles >= 240) return launch_wd<4, 2, 4, true>(p, st);
    return launch_wd<2, 1, 3, true>(p, st);
  }
  if (N % 256 == 0 && big_tiles >= 240) return launch_wd<4, 2, 4>(p, st);
  // 64 x 128 tiles: the software-pipelined loop where more than two workgroups' worth of tiles meet on a CU.  Lab switch PFPP_WD_PF=1, OFF by
  // default: stand-alone it takes 11 % off the 1536-wide GEMM, inside the overlapped training iteration it costs 0.6 % (6.186 / 6.139 / 6.109
  // against 6.129 / 6.072 / 6.091 ms alternating on one box: two waves per SIMD and 48 KB of LDS leave the co-running streams less room) and the
  // compact sampler step does not move (2.92 against 2.93 ms) — profiles/r05zz_ab_wd_pf.txt
  static const bool pf_on = getenv("PFPP_WD_PF") && atoi(getenv("PFPP_WD_PF")) == 1;
  const int64_t small_tiles = ((M + 63) / 64) * (N / 128);
  if (pf_on && small_tiles >= 512 && K >= 7 * 32) return launch_wd<2, 1, 4, false, true>(p, st);
  return launch_wd<2, 1, 3>(p, st);
}

extern "C" int pfpp_gemm_wd(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr,
                            float* out, int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream) {
  return gemm_wd_impl(A, lda, w, bias, residual, ldr, out, ldc, M, N, K, false, stream);
}

extern "C" int pfpp_gemm_wd_f16(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr,
                                float* out, int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream) {
  return gemm_wd_impl(A, lda, w, bias, residual, ldr, out, ldc, M, N, K, true, stream);
}

extern "C" int pfpp_reblock_planes(const pfpp_reblock_job* jobs, int32_t n, pfpp_stream_t stream) {
  PFPP_REQUIRE(jobs && n >= 1 && n <= PFPP_REBLOCK_MAX, "1 .. PFPP_REBLOCK_MAX jobs");
  RbP p;
  p.n = n;
  int64_t wgs = 0;
  for (int i = 0; i < n; ++i) {
    const pfpp_reblock_job& q = jobs[i];
    PFPP_REQUIRE(q.w.hi && q.w.lo && q.fhi && q.flo, "null pointer");
    PFPP_REQUIRE(q.N >= 32 && q.K >= 16 && q.N <= 0x7fffffff && q.K <= 0x7fffffff && q.ldw >= q.K && q.ldw % 8 == 0, "sizes");
    PFPP_SUPPORTED(q.transposed ? (q.N % 64 == 0 && q.K % 64 == 0) : (q.N % 32 == 0 && q.K % 16 == 0), "N / K not a multiple of the block");
    PFPP_REQUIRE(pfpp::aligned16(q.w.hi) && pfpp::aligned16(q.w.lo) && pfpp::aligned16(q.fhi) && pfpp::aligned16(q.flo), "16-byte aligned planes");
    RbJob& r = p.job[i];
    r.hi = (const _Float16*)q.w.hi; r.lo = (const _Float16*)q.w.lo; r.fhi = (_Float16*)q.fhi; r.flo = (_Float16*)q.flo;
    r.N = (int)q.N; r.K = (int)q.K; r.ldw = q.ldw; r.transposed = q.transposed; r.first_wg = (int)wgs;
    wgs += q.transposed ? (q.N / 64) * (q.K / 64) : ((q.N / 32) * (q.K / 16) * 64 + 255) / 256;
  }
  PFPP_REQUIRE(wgs <= 0x7fffffff, "too many workgroups");
  for (int i = n; i < PFPP_REBLOCK_MAX; ++i) p.job[i] = p.job[0];
  hipLaunchKernelGGL(reblock_kernel, dim3((unsigned)wgs), dim3(256), 0, pfpp::as_stream(stream), p);
  return pfpp::check_launch(__func__);
}
