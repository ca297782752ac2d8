// Wave-specialised plane GEMM core: out = (A . W^T) * alpha + bias (+ residual), split-f16 (three v_mfma_f32_32x32x16_f16 per 16-deep step).
//
// Reference: the token-sized linear layers of EncoderLayer.forward (denoiser/model/modules/attention.py:77-90) and their input gradients.
//
// Why (round 6).  Rounds 3-5 measured the 64 x 128 tile at 0.39-0.5 us per 32 contraction values whatever the loop looked like, while its
// parts run in 0.25 (matrix instructions + barrier) and 0.30-0.36 us (loads alone): a wave that waits in the issue of a vector-memory
// instruction issues no matrix instruction either, half-line requests for the activation rows, and one s_barrier per 12 matrix
// instructions.  This kernel takes the memory instructions OUT of the waves that multiply:
//   * workgroup = 8 waves = 4 CONSUMERS (waves 0-3, one per SIMD: ds_read_b128 + v_mfma only, no vector-memory instruction in the K loop)
//     + 4 LOADERS (waves 4-7: global_load_lds_dwordx4 only);
//   * BOTH operands go through a three-stage LDS ring of 64-deep stages (48 KB: activation rows as whole 128-byte lines, XOR-swizzled on
//     the source side; the weight's fragment-blocked planes land as the 1 KB fragments the consumers read back lane-linearly);
//   * no s_barrier in the loop: a stage is handed over through two LDS counters (full: +1 per loader wave whose pieces have landed;
//     empty: +1 per consumer wave that has read its last fragment) - the loaders run up to two stages ahead, across tile boundaries;
//   * persistent: a workgroup walks its tiles (XCD-contiguous, consecutive tiles of a row panel run side by side on one L2), the loaders
//     fill the ring for the next tile while the consumers store the current one - the consumers' vmcnt only ever counts their own stores;
//   * every fragment read of 16-deep step s + 1 sits in the shadow of one matrix instruction of step s, waits are counted (LDS returns
//     in order), so no read latency is exposed in the steady state.
// Arithmetic: lo.hi, hi.lo, hi.hi per 16-deep step, k ascending, one accumulator chain per output, (acc * alpha + bias) + residual -
// gemm_wd_kernel's and gemm_pl_kernel's, bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace pfpp_ws {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct WsP {
  const _Float16 *ah, *al; int64_t lda;      // planes of a_scale * A [M, K], row-major
  const half8 *fh, *fl;                      // fragment-blocked planes of w_scale * W [N, K]: piece ((nb * K/16 + kb) * 64 + lane)
  float alpha;                               // 1 / (a_scale * w_scale)
  const float* bias;                         // [N] or null
  const float* res; int64_t ldr;             // [M, ldr] or null (may alias out)
  float* out; int64_t ldc;
  int M, N, K;
  int tiles;                                 // (M + 63) / 64 * (N / 128)
};

constexpr int BM = 64, BN = 128, BK = 64, D = 3;
constexpr int A_PLANE = BM * BK * 2;                    // 8 KB: 64 rows x 128 bytes
constexpr int W_OFF = 2 * A_PLANE;                      // 16 KB
constexpr int W_UNIT = 2 * (BK / 16) * 1024;            // 8 KB per 32-column unit: [hi: 4 fragments][lo: 4 fragments]
constexpr int STAGE = W_OFF + 4 * W_UNIT;               // 48 KB
constexpr int PATCH_OFF = D * STAGE;                    // consumer epilogue patches: 4 x 2 KB (16 rows x 32 floats)
constexpr int PATCH = 2048;
constexpr int FLAG_OFF = PATCH_OFF + 4 * PATCH;         // full[3] at +0, empty[3] at +16, issued at +32
constexpr int SMEM = FLAG_OFF + 64;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int OFF>
__device__ __forceinline__ half8 lds_rd(uint32_t addr) {
  half8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t flag_rd(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void flag_add(uint32_t addr, int lane) {     // one LDS instruction per wave (lane 0 only)
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}
// spin until the counter at addr reaches need (wave-uniform).  WS_SPIN_LIMIT (lab builds): give up after that many polls, so that a
// protocol bug shows as a wrong result instead of a hung GPU
#ifndef WS_SPIN_LIMIT
#define WS_SPIN_LIMIT 0
#endif
__device__ __forceinline__ void flag_wait(uint32_t addr, uint32_t need) {
  for (int spin = 0;; ++spin) {
    uint32_t v = flag_rd(addr);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= need) break;
    if (WS_SPIN_LIMIT > 0 && spin > WS_SPIN_LIMIT) break;
  }
}

__device__ __forceinline__ void flag_wait_slow(uint32_t addr, uint32_t need) {     // the prefetcher's poll: in no hurry
  for (int spin = 0;; ++spin) {
    uint32_t v = flag_rd(addr);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= need) break;
    if (WS_SPIN_LIMIT > 0 && spin > WS_SPIN_LIMIT) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

// the tiles of workgroup b out of G: XCD x = b % 8 owns a contiguous eighth of the row-major tile list, its workgroups take every wx-th tile of it
struct TileWalk {
  int next, end, step;
  __device__ __forceinline__ TileWalk(int tiles, int b, int G) {
    const int x = b & 7, j = b >> 3;
    const int q = tiles >> 3, r = tiles & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int cnt = q + (x < r ? 1 : 0);
    step = (G - x + 7) >> 3;
    next = start + j;
    end = start + cnt;
  }
  __device__ __forceinline__ int count() const { return next < end ? (end - next + step - 1) / step : 0; }
};

// AB (lab): 1 no matrix instructions, 2 no DMA (the loaders only signal), 4 no fragment reads either, 8 no epilogue stores,
// 16 activation rows fetched without the swizzle, 32 no activation pieces, 64 no weight pieces, 128 rotated K, 256 loaders never wait for a free stage
template <int AB = 0, int PF = 0, int NL = 4>
__device__ __forceinline__ void gemm_ws_body(const WsP& p, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  const uint32_t full0 = lds0 + FLAG_OFF, empty0 = lds0 + FLAG_OFF + 16, issued0 = lds0 + FLAG_OFF + 32;
  if (tid < 16) reinterpret_cast<uint32_t*>(smem + FLAG_OFF)[tid] = 0u;
  __syncthreads();
  const int tiles_n = p.N / BN, KB = p.K / 16, nks = p.K / BK;
  TileWalk walk(p.tiles, blockIdx.x, gridDim.x);
  const int my_tiles = walk.count();
  const int total = my_tiles * nks;                   // stages this workgroup moves through the ring

  constexpr int PA = 16 / NL, PW = 32 / NL, PIECES = PA + PW;     // 1 KB DMA pieces per loader wave and stage: of A, of W
  if (wave == 4 + NL) {
    // ================================================================ prefetcher (a ninth wave, launched only when PF > 0)
    // The activation planes were written by the kernel in front (another XCD: they sit in the memory-side cache or HBM, ~1 us away)
    // and the weights are first-touched by all the workgroups of an XCD in lock-step, while the ring holds two stages in flight: the
    // loaders alone run latency-bound.  This wave touches one dword per 128-byte line of the stage PF positions ahead of the loaders
    // (plain loads into a register nobody reads: their only effect is the L2 fill).  A wave of its own because vector-memory
    // operations of one wave return in order - a slow line in front of the loaders' pieces would hold their hand-over back.
    if constexpr (PF > 0) {
      int h = 0;
      uint32_t sink = 0;     // every load names it read-write: ONE register, live (and never an address) until the final wait
      for (int tile = walk.next; tile < walk.end; tile += walk.step) {
        const int bx = tile / tiles_n, by = tile - bx * tiles_n;
        const int m0 = bx * BM;
        const int64_t grow = m0 + lane < p.M ? m0 + lane : p.M - 1;
        const char* a0 = reinterpret_cast<const char*>(p.ah + grow * p.lda);
        const char* a1 = reinterpret_cast<const char*>(p.al + grow * p.lda);
        const bool do_a = by == 0, do_w = (bx & 3) == 0;     // one workgroup per row panel / every fourth row panel (two or so per XCD)
        // weight lines of a stage: unit lane / 16 .. , 4 instructions x 64 lanes = 256 lines = 4 units x 2 planes x 4 KB
        for (int ks = 0; ks < nks; ++ks, ++h) {
          if (h >= PF) flag_wait_slow(issued0, (uint32_t)(h - PF + 1));
          if (do_a) {
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a0 + (size_t)ks * (BK * 2)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a1 + (size_t)ks * (BK * 2)) : "memory");
          }
          if (do_w) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const int L = n * 64 + lane, c = L >> 6, pl = (L >> 5) & 1;
              const char* w = reinterpret_cast<const char*>((pl ? p.fl : p.fh) + ((size_t)(by * 4 + c) * KB + (size_t)ks * 4) * 64) + (L & 31) * 128;
              asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(w) : "memory");
            }
          }
          asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
    }
    return;
  }
  if (wave >= 4) {
    // ================================================================ loaders
    const int l = wave - 4;
    int g = 0;
    for (int tile = walk.next; tile < walk.end; tile += walk.step) {
      const int bx = tile / tiles_n, by = tile - bx * tiles_n;
      const int m0 = bx * BM;
      const char* src[PIECES];
      uint32_t dst[PIECES];
      // activation pieces 4 l .. 4 l + 3 of 16: plane p / 8, rows 8 (p % 8) + lane / 8; the lane that lands at physical chunk lane % 8 of its
      // row fetches logical chunk (lane % 8) ^ ((row >> 1) & 7) (conflict-free ds_read_b128 of 32 rows x one chunk)
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int q = PA * l + j, pl = q >> 3, row = (q & 7) * 8 + (lane >> 3);
        const int chunk = (AB & 16) ? (lane & 7) : ((lane & 7) ^ ((row >> 1) & 7));
        const int64_t grow = m0 + row < p.M ? m0 + row : p.M - 1;
        src[j] = reinterpret_cast<const char*>((pl ? p.al : p.ah) + grow * p.lda + chunk * 8);
        dst[j] = lds0 + pl * A_PLANE + (q & 7) * 1024;
      }
      // weight pieces 8 l .. 8 l + 7 of 32: unit q / 8, plane (q / 4) % 2, 16-deep step q % 4 of the stage: one fragment = 1 KB contiguous
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        const int q = PW * l + j, c = q >> 3, pl = (q >> 2) & 1, kb = q & 3;
        src[PA + j] = reinterpret_cast<const char*>((pl ? p.fl : p.fh) + ((size_t)(by * 4 + c) * KB + kb) * 64 + lane);
        dst[PA + j] = lds0 + W_OFF + c * W_UNIT + pl * (W_UNIT / 2) + kb * 1024;
      }
      for (int ks = 0; ks < nks; ++ks, ++g) {
        const int st = g % D, round = g / D;
        if (round > 0 && !(AB & 256)) flag_wait(empty0 + 4 * st, 4u * round);      // the four consumers are through the previous tenant of the stage
        if (!(AB & 2)) {
#pragma unroll
          for (int j = 0; j < PIECES; ++j) {
            if ((AB & 32) && j < PA) continue;
            if ((AB & 64) && j >= PA) continue;
            const int kr = (AB & 128) ? (ks + (int)(blockIdx.x >> 3)) % nks : ks;     // lab: every workgroup of an XCD starts at another stage
            const size_t koff = j < PA ? (size_t)kr * (BK * 2) : (size_t)kr * (BK / 16) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void*)(src[j] + koff), (lds_void*)(uintptr_t)(dst[j] + st * STAGE), 16, 0, 0);
          }
        }
        if (PF > 0 && l == 0) flag_add(issued0, lane);              // the prefetcher keeps PF positions ahead of this count
        if (g > 0) {                                                // the stage before this one has landed: hand it over
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AB & 32) ? PW : (AB & 64) ? PA : PIECES) : "memory");
          flag_add(full0 + 4 * ((g - 1) % D), lane);
        }
      }
    }
    if (g > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag_add(full0 + 4 * ((g - 1) % D), lane);
    }
    return;
  }

  // ================================================================== consumers
  const int c = wave;                                 // this wave's 32-column unit of the tile
  const int l31 = lane & 31, lhi = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  uint32_t a_ad[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_ad[s] = lds0 + l31 * 128 + (((2 * s + lhi) ^ sw) << 4);
  const uint32_t w_ad = lds0 + W_OFF + c * W_UNIT + lane * 16;
  const uint32_t patch = lds0 + PATCH_OFF + c * PATCH;

  struct Frag { half8 al0, wh, al1, ah0, wl, ah1; };  // in read order
  Frag F[2];
  f32x16 acc[2];
  uint32_t flagv = 0;

  // the reads of 16-deep step S of the stage at byte offset so, one at a time (i = 0 .. 5)
  auto rd = [&](Frag& f, uint32_t so, auto s_c, auto i_c) {
    constexpr int S = decltype(s_c)::value, i = decltype(i_c)::value;
    if constexpr (AB & 4) return;
    const uint32_t aa = a_ad[S] + so, ww = w_ad + so;
    if constexpr (i == 0) f.al0 = lds_rd<A_PLANE>(aa);
    else if constexpr (i == 1) f.wh = lds_rd<S * 1024>(ww);
    else if constexpr (i == 2) f.al1 = lds_rd<A_PLANE + 4096>(aa);
    else if constexpr (i == 3) f.ah0 = lds_rd<0>(aa);
    else if constexpr (i == 4) f.wl = lds_rd<W_UNIT / 2 + S * 1024>(ww);
    else f.ah1 = lds_rd<4096>(aa);
  };
  auto mfma = [&](const Frag& f, auto i_c) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (AB & 1) {
      if constexpr (i == 0) acc[0][0] += (float)f.al0[0] + (float)f.wh[0];
      else if constexpr (i == 1) acc[1][0] += (float)f.al1[0];
      else if constexpr (i == 2) acc[0][1] += (float)f.ah0[0] + (float)f.wl[0];
      else if constexpr (i == 3) acc[1][1] += (float)f.ah1[0];
      return;
    }
    if constexpr (i == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al0, f.wh, acc[0], 0, 0, 0);
    else if constexpr (i == 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al1, f.wh, acc[1], 0, 0, 0);
    else if constexpr (i == 2) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah0, f.wl, acc[0], 0, 0, 0);
    else if constexpr (i == 3) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah1, f.wl, acc[1], 0, 0, 0);
    else if constexpr (i == 4) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah0, f.wh, acc[0], 0, 0, 0);
    else acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah1, f.wh, acc[1], 0, 0, 0);
  };
  // counted waits.  LDS operations return in issue order, so "read r has returned" == "at most (operations issued behind r) outstanding".
  // W0..W3 = allowed outstanding count in front of matrix instructions 0..3 of a step (instruction 0 needs reads 0-1, 1 needs read 2,
  // 2 needs reads 3-4, 3 needs read 5; 4 and 5 need nothing new).
  auto wait01 = [&](Frag& f, auto n_c) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.al0), "+v"(f.wh) : "n"(decltype(n_c)::value)); };
  auto wait2 = [&](Frag& f, auto n_c) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f.al1) : "n"(decltype(n_c)::value)); };
  auto wait34 = [&](Frag& f, auto n_c) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.ah0), "+v"(f.wl) : "n"(decltype(n_c)::value)); };
  auto wait5 = [&](Frag& f, auto n_c) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f.ah1) : "n"(decltype(n_c)::value)); };
  auto wait_all = [&](Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.al0), "+v"(f.wh), "+v"(f.al1), "+v"(f.ah0), "+v"(f.wl), "+v"(f.ah1));
  };
  using SB = std::integral_constant<int, 0>;
  (void)SB{};
#define WS_PIN() __builtin_amdgcn_sched_barrier(0)
  // one 16-deep step: matrix instruction i of `cur`, then read i of the step behind it into `nxt` (if NEXT)
  //   EXTRA = LDS operations issued at the head of this step behind cur's reads (the flag read of step 2)
  auto step = [&](Frag& cur, Frag& nxt, uint32_t so_next, auto snext_c, auto extra_c, auto next_c) {
    constexpr int X = decltype(extra_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;
    if constexpr (!NEXT) {
      wait_all(cur);
      WS_PIN();
      static_for<6>([&](auto i_c) { mfma(cur, i_c); WS_PIN(); });
      return;
    }
    WS_PIN();
    wait01(cur, std::integral_constant<int, 4 + X>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 0>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 0>{});
    WS_PIN();
    wait2(cur, std::integral_constant<int, 4 + X>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 1>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 1>{});
    WS_PIN();
    wait34(cur, std::integral_constant<int, 3 + X>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 2>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 2>{});
    WS_PIN();
    wait5(cur, std::integral_constant<int, 3 + X>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 3>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 3>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 4>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 4>{});
    WS_PIN(); mfma(cur, std::integral_constant<int, 5>{}); WS_PIN();
    rd(nxt, so_next, snext_c, std::integral_constant<int, 5>{});
    WS_PIN();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using T_ = std::true_type;
  using F_ = std::false_type;

  int g = 0;
  // prologue: the first stage of the first tile
  if (total > 0) {
    flag_wait(full0, (uint32_t)NL);
    static_for<6>([&](auto i_c) { rd(F[0], 0u, I0{}, i_c); });
  }
  const float alpha = p.alpha;
  const int rcol = (lane & 7) * 4, rrow = lane >> 3;
  for (int tile = walk.next; tile < walk.end; tile += walk.step) {
    const int bx = tile / tiles_n, by = tile - bx * tiles_n;
    const int m0 = bx * BM;
    const int col_w = (by * 4 + c) * 32;
    // the residual rows of this wave's outputs, requested now and used in the epilogue (the only vector-memory loads of a consumer);
    // lane -> 4 floats at column 4 (lane % 8) of rows lane / 8 + 8 k of each 16-row half
    f32x4 rq[8];
    if (p.res && !(AB & 8)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + 16 * (i >> 1) + rrow + 8 * (i & 1);
        const int64_t rr = row < p.M ? row : p.M - 1;
        rq[i] = *reinterpret_cast<const f32x4*>(p.res + rr * p.ldr + col_w + rcol);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    for (int ks = 0; ks < nks; ++ks, ++g) {
      const int st = g % D;
      const uint32_t so = st * STAGE;
      // the stage behind this one; past the end the same code runs against this stage again (reads that nobody uses, a counter that
      // is already there): ONE instruction stream for every stage keeps the accumulators where they are
      const bool has_next = g + 1 < total;
      const int stn = has_next ? (g + 1) % D : st;
      const uint32_t son = stn * STAGE;
      const uint32_t need = has_next ? (uint32_t)NL * ((g + 1) / D + 1) : 0u;
      step(F[0], F[1], so, I1{}, I0{}, T_{});
      step(F[1], F[0], so, I2{}, I0{}, T_{});
      // steps 2 and 3 with the hand-over: the next stage's full counter is read at the head of step 2 and looked at at the head of step 3
      flagv = flag_rd(full0 + 4 * stn);
      step(F[0], F[1], so, I3{}, I1{}, T_{});
      WS_PIN();
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(flagv), "+v"(F[1].al0), "+v"(F[1].wh));     // reads 0-1 of step 3 and the flag in front of them
      flag_add(empty0 + 4 * st, lane);                // behind this stage's last read in LDS order: the stage is free once it executes
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)flagv) < need) flag_wait(full0 + 4 * stn, need);
      WS_PIN();
      // step 3, reads of the next stage's step 0 behind its matrix instructions; one more LDS operation (the add) sits behind read 5
      mfma(F[1], I0{}); WS_PIN();
      rd(F[0], son, I0{}, I0{}); WS_PIN();
      wait2(F[1], std::integral_constant<int, 5>{});
      WS_PIN(); mfma(F[1], I1{}); WS_PIN();
      rd(F[0], son, I0{}, I1{}); WS_PIN();
      wait34(F[1], std::integral_constant<int, 4>{});
      WS_PIN(); mfma(F[1], I2{}); WS_PIN();
      rd(F[0], son, I0{}, I2{}); WS_PIN();
      wait5(F[1], std::integral_constant<int, 4>{});
      WS_PIN(); mfma(F[1], I3{}); WS_PIN();
      rd(F[0], son, I0{}, I3{}); WS_PIN();
      mfma(F[1], std::integral_constant<int, 4>{}); WS_PIN();
      rd(F[0], son, I0{}, std::integral_constant<int, 4>{}); WS_PIN();
      mfma(F[1], std::integral_constant<int, 5>{}); WS_PIN();
      rd(F[0], son, I0{}, std::integral_constant<int, 5>{}); WS_PIN();
    }
    // ---- epilogue: (acc * alpha + bias) + residual through the wave's 2 KB patch, 16 rows at a time; 16-byte stores
    if (!(AB & 8)) {
      const float sh = p.bias ? p.bias[col_w + l31] : 0.0f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) {
            const int e = 8 * h + e8;
            const float v = acc[t][e] * alpha;
            const float w = v + sh;
            const int r = (e & 3) + 8 * ((e >> 2) & 1) + 4 * lhi;
            asm volatile("ds_write_b32 %0, %1" ::"v"(patch + r * 128 + l31 * 4), "v"(w) : "memory");
          }
          f32x4 vv[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(vv[k]) : "v"(patch + (rrow + 8 * k) * 128 + rcol * 4) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(F[0].al0), "+v"(F[0].wh), "+v"(F[0].al1), "+v"(F[0].ah0), "+v"(F[0].wl), "+v"(F[0].ah1));
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int row = m0 + t * 32 + 16 * h + rrow + 8 * k;
            if (row < p.M) {
              f32x4 v = vv[k];
              if (p.res) v += rq[4 * t + 2 * h + k];
              *reinterpret_cast<f32x4*>(p.out + (int64_t)row * p.ldc + col_w + rcol) = v;
            }
          }
        }
      }
    }
  }
#undef WS_PIN
}

}  // namespace pfpp_ws
