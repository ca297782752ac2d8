// split-f16 x3 GEMM, direct-to-LDS ring variant (arithmetic: see gemm.hip).
//
// The register-staged kernel keeps one K-tile in flight per workgroup; at the f16 MFMA rate a
// K-tile lasts ~770 cycles per SIMD, less than an L2 / Infinity-Cache round trip, so it runs
// latency-bound (rocprofv3: 47 % of its wave cycles in SQ_WAIT_ANY, matrix pipe 27 % busy).  Here
// every global load is an LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB of LDS per wave
// instruction, no VGPRs) into a ring of NS stages; NS-1 K-tiles (96 KB at NS = 4) are in flight
// per CU, waits are counted (each thread issues 8 DMAs per stage and they retire in order:
// s_waitcnt vmcnt(8*(NS-2))) and there is one raw s_barrier per K-tile.
//
// An LDS-DMA image is lane-linear, so bank conflicts are removed by permuting the 16-byte chunks
// of a row on the SOURCE side (chunk ^= f(row)) and reading with the same XOR (cdna guide rule
// 21): fp32 rows are 128 B (f = (row>>1)&7), f16 plane rows are 64 B (f = (row>>2)&3).
//
// Operands: W always as pre-split f16 planes.  A either as pre-split planes (APRE: produced by the
// LayerNorm / attention kernels and the GEGLU epilogue — the loop is then DMA + ds_read + MFMA
// only) or as fp32 (split into hi/lo when a wave reads its fragments; measured issue-bound).
//
// Fragment reads are inline-asm ds_read_b128: once an LDS-DMA is in flight hipcc drains vmcnt(0)
// in front of every ds_read it can see, which would serialise the ring; asm reads are invisible
// to that bookkeeping and are waited for explicitly, naming every destination (cdna guide §5.7).
// This file holds a single __shared__ object on purpose.
#include "gemm_common.h"

namespace pfpp_gemm_detail {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
constexpr int BK = 32;

__device__ __forceinline__ void glds16(const void* gsrc, char* ldst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(uintptr_t)ldst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

__device__ __forceinline__ v4f lds_rd_f4(uint32_t addr) {
  v4f v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ half8 lds_rd_h8(uint32_t addr) {
  half8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

__device__ __forceinline__ void split8(const v4f v0, const v4f v1, half8& hi, half8& lo) {
  const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)x[e];
    hi[e] = h;
    lo[e] = (_Float16)(x[e] - (float)h);
  }
}

template <int NS, bool APRE>
__global__ __launch_bounds__(256) void gemm_f16x3_ring_kernel(const GemmP p) {
  constexpr int BM = 128, BN = 128;
  constexpr int A_BYTES = BM * BK * 4;         // 16 KB: fp32 rows of 128 B, or two f16 planes of 8 KB
  constexpr int PLANE = 128 * BK * 2;          // 8 KB f16 plane, rows of 64 B
  constexpr int STAGE_BYTES = A_BYTES + 2 * PLANE;
  extern __shared__ __align__(16) char ring_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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

  // ---- DMA sources (per lane; the k offset is added per stage) --------------------------------------
  const char* a_src[4];     // fp32: 4 x (8 rows x 128 B); planes: hi q=0,1 then lo q=0,1 (16 rows x 64 B)
  const char* w_src[4];     // hi q=0,1 then lo q=0,1
  {
    // f16 plane pieces: row = wave*32 + q*16 + lane/4, 16-byte chunk = (lane%4) ^ ((row>>2)&3)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = wave * 32 + q * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const int64_t woff = w_offz + (int64_t)min(n0 + row, p.N - 1) * p.ldw + chunk * 8;
      w_src[q] = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(p.Whi) + woff);
      w_src[2 + q] = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(p.Wlo) + woff);
      if constexpr (APRE) {
        const int64_t aoff = a_offz + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8;
        a_src[q] = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(p.Ahi) + aoff);
        a_src[2 + q] = reinterpret_cast<const char*>(reinterpret_cast<const _Float16*>(p.Alo) + aoff);
      }
    }
    if constexpr (!APRE) {
      // fp32 pieces: row = wave*32 + q*8 + lane/8, chunk = (lane%8) ^ ((row>>1)&7)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[q] = reinterpret_cast<const char*>(p.A + a_offz + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 4);
      }
    }
  }
  auto issue = [&](int kt, int slot) {
    char* st = ring_smem + slot * STAGE_BYTES;
    if constexpr (APRE) {
      const int kb = kt * BK * 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        glds16(a_src[q] + kb, st + (wave * 32 + q * 16) * 64);
        glds16(a_src[2 + q] + kb, st + PLANE + (wave * 32 + q * 16) * 64);
      }
    } else {
      const int kb = kt * BK * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(a_src[q] + kb, st + (wave * 32 + q * 8) * 128);
    }
    const int kbw = kt * BK * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      glds16(w_src[q] + kbw, st + A_BYTES + (wave * 32 + q * 16) * 64);
      glds16(w_src[2 + q] + kbw, st + A_BYTES + PLANE + (wave * 32 + q * 16) * 64);
    }
  };

  f32x16 accM[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) accM[i][j][e] = 0.0f;

  // ---- fragment addressing ---------------------------------------------------------------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)ring_smem;
  uint32_t a_addr[2], w_addr[2];
  int a_sw[2], w_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + l31;
    a_addr[i] = lds0 + (APRE ? r * 64 : r * 128);
    a_sw[i] = APRE ? (r >> 2) & 3 : (r >> 1) & 7;
    const int rr = wn * 64 + i * 32 + l31;
    w_addr[i] = lds0 + A_BYTES + rr * 64;
    w_sw[i] = (rr >> 2) & 3;
  }
  struct Frags { v4f a0[2], a1[2]; half8 ah[2], al[2], bh[2], bl[2]; };
  auto read_frags = [&](Frags& f, int slot, int ks) {
    const uint32_t st = slot * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (APRE) {
        const int c = (ks * 2 + lhi) ^ a_sw[i];
        f.ah[i] = lds_rd_h8(a_addr[i] + st + c * 16);
        f.al[i] = lds_rd_h8(a_addr[i] + st + PLANE + c * 16);
      } else {
        const int c0 = (ks * 4 + lhi * 2) ^ a_sw[i];
        f.a0[i] = lds_rd_f4(a_addr[i] + st + c0 * 16);
        f.a1[i] = lds_rd_f4(a_addr[i] + st + (c0 ^ 1) * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (ks * 2 + lhi) ^ w_sw[j];
      f.bh[j] = lds_rd_h8(w_addr[j] + st + c * 16);
      f.bl[j] = lds_rd_h8(w_addr[j] + st + PLANE + c * 16);
    }
  };
  auto wait_frags = [&](Frags& f) {
    if constexpr (APRE) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.bh[0]), "+v"(f.bh[1]),
                     "+v"(f.bl[0]), "+v"(f.bl[1])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.a0[0]), "+v"(f.a0[1]), "+v"(f.a1[0]), "+v"(f.a1[1]), "+v"(f.bh[0]), "+v"(f.bh[1]),
                     "+v"(f.bl[0]), "+v"(f.bl[1])
                   :
                   : "memory");
    }
  };
  auto mma = [&](Frags& f) {
    if constexpr (!APRE) {
#pragma unroll
      for (int i = 0; i < 2; ++i) split8(f.a0[i], f.a1[i], f.ah[i], f.al[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], accM[i][j], 0, 0, 0);
        accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], accM[i][j], 0, 0, 0);
        accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], accM[i][j], 0, 0, 0);
      }
  };

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; up to NS-2 younger stages may stay in flight
    const int younger = min(NS - 2, nk - 1 - kt);
    if (younger >= 2) wait_vmcnt<16>();
    else if (younger == 1) wait_vmcnt<8>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // every wave's part of stage kt is in LDS; stage kt-1 is no longer read
    if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
    const int slot = kt % NS;
    Frags f0, f1;
    read_frags(f0, slot, 0);
    wait_frags(f0);
    read_frags(f1, slot, 1);     // in flight under the first k-step's MFMAs
    mma(f0);
    wait_frags(f1);
    mma(f1);
  }

  epilogue<2, 2>(p, accM, m0 + wm * 64, n0 + wn * 64, n0, wn, lane, c_off, v_off);
}

template <bool APRE>
static int launch_ring(const GemmP& p0, int batch, hipStream_t st, int group_m) {
  constexpr int NS = 4;
  constexpr size_t smem = (size_t)NS * (128 * 32 * 4 + 2 * 128 * 32 * 2);
  static bool attr_set = false;
  auto kern = gemm_f16x3_ring_kernel<NS, APRE>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmP p = p0;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 127) / 128;
  p.group_m = p.tiles_n > 1 ? group_m : 0;
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  return pfpp::check_launch("pfpp_gemm");
}

int launch_f16x3_ring(const GemmP& p, int batch, hipStream_t st, int group_m) {
  return p.Ahi ? launch_ring<true>(p, batch, st, group_m) : launch_ring<false>(p, batch, st, group_m);
}

}  // namespace pfpp_gemm_detail
