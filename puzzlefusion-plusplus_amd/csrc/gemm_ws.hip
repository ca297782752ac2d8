// split-f16 x3 GEMM, wave-specialised variant (arithmetic and epilogues: see gemm.hip).
//
// Counter analysis of the register-staged kernel (profiles/r01_gemm_f16x3_pmc_*): 47 % of its wave
// cycles wait on vmcnt / the barrier — every wave both fetches and multiplies, so a K-tile's loads
// have only one ~770-cycle MFMA phase to come back; the LDS-DMA ring removes the waiting but pays
// ~100 issue cycles per DMA piece in every wave.  Here the roles are split:
//
//   waves 0-3  CONSUMERS  one per SIMD, each owns a 64x64 quadrant of the 128x128 tile: ds_read_b128
//                         fragments + 24 MFMAs per K-tile, nothing else;
//   waves 4-7  LOADERS    one per SIMD, each owns 32 rows of A and of W: plain global_load_dwordx4
//                         into THREE register sets (three K-tiles = 96 KB per CU in flight, no LDS-DMA),
//                         hi/lo split on the VALU (which idles under the consumer's MFMAs: the two
//                         pipes of a SIMD run concurrently), ds_write into a 2-stage LDS buffer.
//
// One s_barrier per K-tile keeps the roles in lock step: during iteration i the loaders fill stage
// i&1 with tile i while the consumers multiply tile i-1 out of stage (i-1)&1.
#include "gemm_common.h"

namespace pfpp_gemm_detail {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr int WS_BK = 32;
constexpr int WS_LDH = WS_BK + 8;                 // 40 halfs = 80 B rows: conflict-free b128 reads
constexpr int WS_PLANE = 128 * WS_LDH;            // halfs
constexpr int WS_STAGE = 4 * WS_PLANE;            // Ahi | Alo | Whi | Wlo

__device__ __forceinline__ float4 ws_load_k4(const float* row, int k, int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) {
    v = *reinterpret_cast<const float4*>(row + k);
    if (k + 4 > K) {
      if (k + 1 >= K) v.y = 0.f;
      if (k + 2 >= K) v.z = 0.f;
      v.w = 0.f;
    }
  }
  return v;
}

// the LDS writes of the loaders must have landed before the barrier releases the consumers; the
// "memory" clobber also keeps the compiler from moving LDS accesses across (s_barrier is IntrNoMem)
__device__ __forceinline__ void ws_barrier_after_writes() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void ws_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct WsRegs {
  float4 a[4];
  uint4 wh[2], wl[2];
};

__global__ __launch_bounds__(512) void gemm_f16x3_ws_kernel(const GemmP p) {
  extern __shared__ __align__(16) _Float16 ws_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tile = remap_tile(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(p, tile, tm, tn);
  const int m0 = tm * 128, n0 = tn * 128;
  const int z = blockIdx.z;
  const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
  const int nk = (p.K + WS_BK - 1) / WS_BK;

  if (wave >= 4) {
    // ======================================= LOADER =============================================
    const int lw = wave - 4;
    const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const int64_t w_off = z0 * p.sW0 + z1 * p.sW1;
    const int a_r = lw * 32 + (lane >> 3), a_c4 = lane & 7;       // + 8*it
    const int h_r = lw * 32 + (lane >> 2), h_c8 = lane & 3;       // + 16*it
    const float* a_ptr[4];
    const _Float16* wh_ptr[2];
    const _Float16* wl_ptr[2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
      a_ptr[it] = A + (int64_t)min(m0 + a_r + 8 * it, p.M - 1) * p.lda + a_c4 * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t off = w_off + (int64_t)min(n0 + h_r + 16 * it, p.N - 1) * p.ldw + h_c8 * 8;
      wh_ptr[it] = reinterpret_cast<const _Float16*>(p.Whi) + off;
      wl_ptr[it] = reinterpret_cast<const _Float16*>(p.Wlo) + off;
    }
    auto load = [&](WsRegs& r, int kt) {
      const int k0 = kt * WS_BK;
      if (k0 + WS_BK <= p.K) {
#pragma unroll
        for (int it = 0; it < 4; ++it) r.a[it] = *reinterpret_cast<const float4*>(a_ptr[it] + k0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          r.wh[it] = *reinterpret_cast<const uint4*>(wh_ptr[it] + k0);
          r.wl[it] = *reinterpret_cast<const uint4*>(wl_ptr[it] + k0);
        }
      } else {   // ragged last tile: A masked element-wise, W planes are zero padded to 8 halfs
#pragma unroll
        for (int it = 0; it < 4; ++it) r.a[it] = ws_load_k4(a_ptr[it] - a_c4 * 4, k0 + a_c4 * 4, p.K);
        const bool ok = k0 + h_c8 * 8 < (int)p.ldw;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          r.wh[it] = ok ? *reinterpret_cast<const uint4*>(wh_ptr[it] + k0) : make_uint4(0, 0, 0, 0);
          r.wl[it] = ok ? *reinterpret_cast<const uint4*>(wl_ptr[it] + k0) : make_uint4(0, 0, 0, 0);
        }
      }
    };
    auto store = [&](const WsRegs& r, int stage) {
      _Float16* st = ws_smem + stage * WS_STAGE;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float x[4] = {r.a[it].x, r.a[it].y, r.a[it].z, r.a[it].w};
        half4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)x[e];
          hi[e] = h;
          lo[e] = (_Float16)(x[e] - (float)h);
        }
        const int off = (a_r + 8 * it) * WS_LDH + a_c4 * 4;
        *reinterpret_cast<half4*>(st + off) = hi;
        *reinterpret_cast<half4*>(st + WS_PLANE + off) = lo;
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int off = (h_r + 16 * it) * WS_LDH + h_c8 * 8;
        *reinterpret_cast<uint4*>(st + 2 * WS_PLANE + off) = r.wh[it];
        *reinterpret_cast<uint4*>(st + 3 * WS_PLANE + off) = r.wl[it];
      }
    };
    WsRegs r0, r1, r2;
    if (0 < nk) load(r0, 0);
    if (1 < nk) load(r1, 1);
    if (2 < nk) load(r2, 2);
    // iteration i: write tile i (register set i%3) into stage i&1, refill the set with tile i+3
    for (int i = 0; i <= nk; i += 3) {
      if (i <= nk) {
        if (i < nk) { store(r0, i & 1); if (i + 3 < nk) load(r0, i + 3); }
        ws_barrier_after_writes();
      }
      if (i + 1 <= nk) {
        if (i + 1 < nk) { store(r1, (i + 1) & 1); if (i + 4 < nk) load(r1, i + 4); }
        ws_barrier_after_writes();
      }
      if (i + 2 <= nk) {
        if (i + 2 < nk) { store(r2, (i + 2) & 1); if (i + 5 < nk) load(r2, i + 5); }
        ws_barrier_after_writes();
      }
    }
    return;
  }

  // ========================================= CONSUMER ==============================================
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  f32x16 accM[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) accM[i][j][e] = 0.0f;
  const int a_base = (wm * 64 + l31) * WS_LDH + lhi * 8;
  const int w_base = 2 * WS_PLANE + (wn * 64 + l31) * WS_LDH + lhi * 8;

  for (int i = 0; i <= nk; ++i) {
    if (i >= 1) {
      const _Float16* st = ws_smem + ((i - 1) & 1) * WS_STAGE;
#pragma unroll
      for (int ks = 0; ks < WS_BK / 16; ++ks) {
        half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[m] = *reinterpret_cast<const half8*>(st + a_base + m * 32 * WS_LDH + ks * 16);
          al[m] = *reinterpret_cast<const half8*>(st + WS_PLANE + a_base + m * 32 * WS_LDH + ks * 16);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          bh[n] = *reinterpret_cast<const half8*>(st + w_base + n * 32 * WS_LDH + ks * 16);
          bl[n] = *reinterpret_cast<const half8*>(st + WS_PLANE + w_base + n * 32 * WS_LDH + ks * 16);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            accM[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], accM[m][n], 0, 0, 0);
            accM[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], accM[m][n], 0, 0, 0);
            accM[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], accM[m][n], 0, 0, 0);
          }
      }
    }
    ws_barrier();
  }

  const int64_t c_off = z0 * p.sC0 + z1 * p.sC1;
  const int64_t v_off = z0 * p.sV0 + z1 * p.sV1;
  epilogue<2, 2>(p, accM, m0 + wm * 64, n0 + wn * 64, n0, wn, lane, c_off, v_off);
}

int launch_f16x3_ws(const GemmP& p0, int batch, hipStream_t st, int group_m) {
  constexpr size_t smem = (size_t)2 * WS_STAGE * sizeof(_Float16);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_ws_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmP p = p0;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 127) / 128;
  p.group_m = p.tiles_n > 1 ? group_m : 0;
  const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)batch);
  hipLaunchKernelGGL(gemm_f16x3_ws_kernel, grid, dim3(512), smem, st, p);
  return pfpp::check_launch("pfpp_gemm");
}

}  // namespace pfpp_gemm_detail
