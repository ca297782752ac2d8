/*
 * pfpp.h — C ABI of libpfpp_hip.so: the MI355X (gfx950) kernels behind the
 * denoise-and-verify hot path of PuzzleFusion++.
 *
 * The reference has no native code on this path: every function below replaces
 * a group of stock-PyTorch / third-party ops.  Each entry cites the reference
 * lines it replaces (paths relative to the reference checkout).  The only
 * native precedent in the reference tree is Jigsaw_matching/utils/chamfer/cuda/
 * chamfer.cpp:8-23 (contiguity checks on the host side, raw pointers + sizes
 * into the launcher); this header follows the same split: the Python wrappers
 * validate dtype / contiguity / shapes, the C side takes plain pointers.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - all tensors are dense row-major; "ld*" = leading dimension in elements;
 *   - fp32 everywhere (the reference runs precision 32,
 *     config/denoiser/global_config.yaml:39); indices are int32 at this
 *     boundary (the Python mirror widens to int64 where the reference API
 *     returns int64);
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued,
 *     never synchronised; nothing is allocated or freed by the library;
 *   - return value: PFPP_OK (0) or a negative PFPP_E* code.  Nothing throws
 *     across the ABI.  pfpp_last_error() returns a static string describing
 *     the most recent failure on the calling thread.
 */
#ifndef PFPP_H_
#define PFPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFPP_OK 0
#define PFPP_EINVAL (-1)     /* bad size / alignment / null pointer            */
#define PFPP_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define PFPP_EHIP (-3)       /* hipGetLastError() != hipSuccess after launch   */

typedef void* pfpp_stream_t;

/* ---- library ------------------------------------------------------------ */
#define PFPP_ABI_VERSION 2
int pfpp_version(void);                 /* ABI version = PFPP_ABI_VERSION (2: pfpp_build_info, per-thread attention mode) */
/* "abi=2;arch=gfx950;fma_mix_insts=off;packed_fp32_ops=off;chain_prio=3": the code-generation switches this binary was compiled
 * with.  Two of them are CORRECTNESS switches: a binding must refuse a library that does not say "off" for both
 * (pfpp_hip/_lib.py does; PFPP_PACKED_FP32=1 is the lab override for the second) */
const char* pfpp_build_info(void);
const char* pfpp_last_error(void);
/* sizeof(struct pfpp_<name>) as this library was compiled ("gemm_planes_args", "tlayers_args", ...), -1 for an unknown name: a binding
 * that mirrors the structs (pfpp_hip/_lib.py) checks its own layout against it when it loads, so a stale mirror fails loudly    */
int64_t pfpp_abi_sizeof(const char* name);
/* number of compute units of the current device (for grid sizing in hosts) */
int pfpp_device_cu_count(void);
/* arithmetic of the attention FORWARD kernels for launches issued by the CALLING THREAD (thread-local like pfpp_last_error: the
 * library keeps no process-global mutable state) (pfpp_attn_dense*, pfpp_attn_blockdiag*; diffusers Attention,
 * attention.py:46-72 via denoiser_transformer.py:46-72): -1 = each kernel's default (split-f16 unless its PFPP_ATTN_* environment
 * switch says otherwise), 0 = exact fp32 matrix instructions (the range fallback of the sampler loops, pfpp_hip.ops.exact_fp32),
 * 1 = split-f16, 2 = single-pass fp16 (perf mode of BASELINE configs[4]; never a parity mode). */
int pfpp_set_attention_mode(int mode);
int pfpp_get_attention_mode(void);

/* ---- a1: SE(3) rotate + valid-fragment gather ----------------------------
 * Denoiser._apply_rots (puzzlefusion_plusplus/denoiser/model/denoiser.py:55-63,
 * = auto_aggl.py:70-78) followed by the boolean gather denoiser.py:69.
 *   q = pose[slot,3:7] / ||q||   (w,x,y,z);  out = (q * (0,p) * conj(q)).xyz
 * evaluated in exactly pytorch3d's quaternion_raw_multiply operation order with
 * no FMA contraction, so the rotated coordinates are bit-identical to the CPU
 * path (FPS downstream is an argmax chain and must not fork).
 *   part_pcs [n_slots, N, 3], pose [n_slots, 7], slot [F] (flattened b*P+p of
 *   each valid fragment, ascending) -> out [F, N, 3]                          */
int pfpp_se3_rotate_gather(const float* part_pcs, const float* pose,
                           const int32_t* slot, float* out,
                           int64_t F, int64_t N, pfpp_stream_t stream);

/* ---- a19: pose apply  R(q^)p + t  ----------------------------------------
 * utils/node_merge_utils.py:43-53 get_final_pose_pts (normalise=1) and the
 * per-part body of get_final_pose_pts_dynamic :16-41 (normalise=0: pytorch3d
 * quaternion_apply does not normalise).  pts [n, N, 3], pose [n,7]=(t, q).
 * `scale` (may be NULL) multiplies the points first (auto_aggl.py:160-162). */
int pfpp_pose_apply(const float* pts, const float* pose, const float* scale,
                    float* out, int64_t n, int64_t N, int normalise,
                    pfpp_stream_t stream);

/* ---- a2: farthest point sampling -----------------------------------------
 * torch_cluster.fps(random_start=False) as called at utils/pn2_utils.py:131-137
 * + the centroid gather index_points :139.  Start index 0, d = (dx*dx + dy*dy)
 * + dz*dz in fp32 without FMA, running min, next = first argmax.
 *   xyz [F, N, 3] -> idx [F, S] (local indices), new_xyz [F, S, 3]
 * Supported: 1 <= S <= N <= 4096.                                           */
int pfpp_fps(const float* xyz, int32_t* idx, float* new_xyz,
             int64_t F, int64_t N, int64_t S, pfpp_stream_t stream);

/* ---- a3: ball query ---------------------------------------------------------
 * square_distance + query_ball_point, utils/pn2_utils.py:21-42, 92-112.
 * d = ((-2*dot) + |c|^2) + |p|^2 with dot = fma(c2,p2, fma(c1,p1, c0*p0))
 * (what the CPU BLAS K=3 matmul evaluates) and squared norms (x*x+y*y)+z*z;
 * a point is kept unless d > r2; the first `nsample` kept indices in index
 * order are returned, padded with the first kept index.
 *   xyz [F,N,3], new_xyz [F,S,3] -> idx [F,S,nsample]; nsample <= 64.       */
int pfpp_ball_query(const float* xyz, const float* new_xyz, int32_t* idx,
                    int64_t F, int64_t N, int64_t S, int64_t nsample,
                    float r2, pfpp_stream_t stream);

/* ---- a2 + a3 for the three set-abstraction levels at once ------------------
 * PN2.encode (vqvae/model/modules/pn2.py:57-68) runs sample_and_group (utils/pn2_utils.py:127-151) three times; the
 * sampling part (farthest_point_sample :131-134 + query_ball_point :92-111) depends on the coordinates only.  One
 * launch, one workgroup per fragment: level l samples S_l points out of the S_(l-1) centroids of the level above (level 0: the
 * N input points) and ball-queries them with radius^2 r2_l.  Outputs exactly what pfpp_fps / pfpp_ball_query return for
 * each level (bit-identical).  N <= 2048, levels[0].S <= 256, levels[1].S <= 128, nsample <= 64.                        */
typedef struct pfpp_sample_level {
  int64_t S, nsample;
  float r2;
  int32_t* fps_idx;    /* [F, S] or NULL */
  float* new_xyz;      /* [F, S, 3] */
  int32_t* ball_idx;   /* [F, S, nsample] */
} pfpp_sample_level;
int pfpp_sample_levels(const float* xyz, int64_t F, int64_t N, const pfpp_sample_level* levels /* [3] */, pfpp_stream_t stream);

/* ---- a4: grouping --------------------------------------------------------
 * index_points x3 + concat, utils/pn2_utils.py:139-146, written channels-last
 * as the A operand of the first set-abstraction GEMM:
 *   row (f,s,j) = [ feats[f, idx, 0:D] | xyz[f,idx]-new_xyz[f,s] | 0-pad ]
 * (features first so the D-wide copy is 16-byte aligned; the packed conv
 * weight has its input columns permuted the same way).
 *   feats [F,N,D] or NULL (D=0), out [F*S*ns, ldo], ldo >= D+3, ldo % 4 == 0 */
int pfpp_group_gather(const float* xyz, const float* new_xyz,
                      const float* feats, const int32_t* idx, float* out,
                      int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D,
                      int64_t ldo, pfpp_stream_t stream);

/* ---- GEMM: C = epilogue(A . op(W)) on fp32 MFMA ----------------------------
 * The one contraction kernel behind a5 (1x1 conv + BN + ReLU + max over
 * nsample, utils/pn2_utils.py:209-214), a6 (conv6, pn2.py:65), a9 (Linear
 * 148->512 / 147->512, denoiser_transformer.py:131-134), a11 (AdaLN linear,
 * attention.py:22), a12-a14 (to_q/k/v, QK^T, PV, to_out, GEGLU feed-forward,
 * attention.py:77-90), a15 (output heads, denoiser_transformer.py:138-147)
 * and a18 (verifier, verifier_transformer.py:42-66).
 * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate.
 *
 *   A [M,K] row-major (lda);  W is [N,K] row-major (w_kmajor=0, the layout of
 *   torch Linear / 1x1-conv weights) or [K,N] row-major (w_kmajor=1);
 *   C [M,N] (ldc) — or [M/pool, N] when pool > 0.
 *   lda, ldw multiples of 4, A/W 16-byte aligned.
 * Batched: blockIdx.z = z; operand X is offset by (z / zdiv)*sX0 + (z % zdiv)*sX1.
 * Epilogue order: v = acc; v = v*scale[n] + shift[n] (if scale) else v += bias[n]
 * (if bias); activation; v += R[m,n] (if residual); store or max-pool.       */
enum {
  PFPP_ACT_NONE = 0,
  PFPP_ACT_RELU = 1,
  PFPP_ACT_SILU = 2,
  PFPP_ACT_GELU = 3,  /* exact erf GELU */
  PFPP_ACT_GEGLU = 4  /* C[m,j] = u * gelu(g): u,g = value/gate columns; W packed
                         as 32-column value block followed by its 32-column gate
                         block (see pfpp_hip.pack_geglu); C has N/2 columns   */
};

/* arithmetic of the contraction (see csrc/gemm.hip):
 *   PFPP_GEMM_F32   v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s roofline)
 *   PFPP_GEMM_F16X3 split-f16: x = hi + lo (hi = f16(x), lo = f16(x - hi)), A.W = hi.hi + hi.lo + lo.hi on
 *                   v_mfma_f32_32x32x16_f16 — fp32-grade error (dropped term 2^-22 relative; elements
 *                   below 2^-3 carry an absolute error <= 3e-8) at up to 16/3 of the fp32-MFMA rate;
 *                   needs |x| < 65504; not available with w_kmajor.
 *                   W may be handed over pre-split (w_hi/w_lo: fp16 [N, ldw] planes, ldw = K rounded
 *                   up to 8 and zero padded) or as fp32 (split on the fly). */
enum { PFPP_GEMM_F32 = 0, PFPP_GEMM_F16X3 = 1, PFPP_GEMM_F16 = 2 };

/*   PFPP_GEMM_F16   single-pass fp16 on pre-split operands (a_hi / w_hi planes only, ONE v_mfma_f32_32x32x16_f16 per product,
 *                   fp32 accumulate): the perf mode of BASELINE.json configs[4].  ~1e-3 relative error: not the parity mode. */
typedef struct pfpp_gemm_args {
  const float* A; const float* W; float* C;
  const void* w_hi; const void* w_lo;   /* pre-split fp16 planes of W, or NULL */
  const void* a_hi; const void* a_lo;   /* pre-split fp16 planes of A [M, lda] (then A may be NULL; needs
                                           w_hi/w_lo, K % 32 == 0, lda % 8 == 0), or NULL              */
  void* c_hi; void* c_lo;               /* if set: write the result as split planes [.., ldc] instead of C */
  const float* bias;      /* [N] or NULL */
  const float* scale;     /* [N] or NULL (then shift must be set) */
  const float* shift;     /* [N] */
  const float* residual;  /* [M, ldr] or NULL; batched with C's strides */
  int64_t M, N, K;
  int64_t lda, ldw, ldc, ldr;
  int32_t w_kmajor;       /* 0: W[N,K]; 1: W[K,N] */
  int32_t act;            /* PFPP_ACT_* */
  int32_t pool;           /* 0, 32 or 64: max over groups of `pool` rows */
  int32_t batch, zdiv;    /* batch >= 1; zdiv >= 1 */
  int32_t precision;      /* PFPP_GEMM_* */
  int64_t sA0, sA1, sW0, sW1, sC0, sC1;
  int64_t sV0, sV1;       /* batch strides of bias / scale / shift */
  float alpha;            /* acc *= alpha before the epilogue (1.0 = off) */
  /* train-mode BatchNorm fusion (set-abstraction MLPs with the encoder in .train(), see pfpp_bn_finalize):
   * a_mul/a_add [K]: A is read as relu(A*a_mul[k] + a_add[k]);  stats: fp64 [stats_copies][2][N], receives
   * (atomically) the column sums and sums of squares of the bias-added result;  c_min: with pool > 0 the
   * per-group minimum [M/pool, ldc] next to the maximum in C.  All NULL/0 = off.                        */
  const float* a_mul; const float* a_add;
  double* stats; int32_t stats_copies;
  float* c_min;
  /* optional split-K workspace (skinny GEMMs: few output tiles, long K — the B = 1 sampler step): when lent, the
   * library may cut K over several workgroups per tile; partials are parked in split_ws (>= split_ws_bytes), the
   * last workgroup of a tile (tickets in split_cnt, split_cnt_len ints, all zero between launches) adds them in chunk
   * order and runs the epilogue — results are deterministic.  Must not be shared by launches that may overlap.  */
  float* split_ws; int64_t split_ws_bytes;
  int32_t* split_cnt; int64_t split_cnt_len;
  /* fused grouping (a4 + a5, utils/pn2_utils.py:127-151 followed by the first 1x1 convolution, :210-213): when
   * gather_idx is set, row r = (f*S + s)*ns + j of the A operand is never materialised — it is read in place as
   *   [ A[f*gather_N + gather_idx[r], 0:D] | gather_xyz[f, idx] - gather_ctr[f*S + s] | 0 ],  D = lda, K = D + 4
   * i.e. exactly what pfpp_group_gather writes (A = the level's input features [F*N, D], unused when D = 0).
   * Needs the f16x3 path with pre-split W, D % 32 == 0, batch 1, no a_mul.  NULL = off.                      */
  const int32_t* gather_idx; const float* gather_xyz; const float* gather_ctr;
  int32_t gather_N, gather_S, gather_ns;
} pfpp_gemm_args;

int pfpp_gemm(const pfpp_gemm_args* args, pfpp_stream_t stream);

/* Optional split-f16 copy of a kernel's result, the operand format of pfpp_gemm_planes: hi = f16(scale * v),
 * lo = f16(scale * v - hi), same indexing as the fp32 result.  scale is a power of two (gradients are lifted into the
 * normal fp16 range; the consuming GEMM divides it out through alpha).  hi / lo: 8-byte aligned fp16 buffers.       */
typedef struct pfpp_planes { void* hi; void* lo; float scale; } pfpp_planes;

/* Plane-producing forms of the training kernels (a17): same arithmetic as the entry points they extend (cited
 * there), with the result additionally — or, where the fp32 pointer may be NULL, only — written as pfpp_planes for the
 * following pfpp_gemm_planes launches: no separate conversion pass over the activations / gradients.
 *   pfpp_split_planes          planes of an fp32 tensor (n % 4 == 0)
 *   pfpp_colsum_planes         out[c] += out_scale * sum_r (hi + lo)[r, c]   (bias gradients of dY given as planes)
 *   pfpp_geglu_p / _bwd_p      pfpp_geglu / pfpp_geglu_bwd; u (dz) may be NULL
 *   pfpp_dropout_layernorm_p   pfpp_dropout_layernorm; n_out may be NULL
 *   pfpp_layernorm_bwd_p       pfpp_layernorm_bwd(_dropout): `dropout` selects the fused dropout of the backward chain
 *                              (drop_out may then be NULL); ret_planes = planes of the value the chain continues with
 *                              (dropped-out gradient, or the updated dx without dropout), dx_planes = planes of the updated dx
 *   pfpp_attn_dense_train_p    pfpp_attn_dense_train; out may be NULL
 *   pfpp_attn_dense_bwd_p, pfpp_attn_blockdiag_bwd_p    dqkv may be NULL (split-f16 / matrix-core kernels only)      */
int pfpp_split_planes(const float* x, int64_t n, const pfpp_planes* planes, pfpp_stream_t stream);
int pfpp_colsum_planes(const void* hi, const void* lo, float* out, int64_t rows, int64_t cols, int64_t ld,
                       float out_scale, pfpp_stream_t stream);
int pfpp_geglu_p(const float* z, float* u, int64_t rows, int64_t inner, float p, uint64_t seed, uint32_t site,
                 const pfpp_planes* u_planes, pfpp_stream_t stream);
int pfpp_geglu_bwd_p(const float* z, const float* du, float* dz, int64_t rows, int64_t inner, float p, uint64_t seed,
                     uint32_t site, const pfpp_planes* dz_planes, pfpp_stream_t stream);
int pfpp_dropout_layernorm_p(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                             int64_t ld_mod, const float* gamma, const float* beta, const int32_t* group_batch,
                             int64_t group_rows, int64_t rows_per_batch, int64_t rows, int64_t C, float eps, float p,
                             uint64_t seed, uint32_t site, const pfpp_planes* n_planes, pfpp_stream_t stream);
int pfpp_layernorm_bwd_p(const float* x, const float* dy, const float* mod, int64_t ld_mod, const float* gamma,
                         const int32_t* group_batch, int64_t group_rows, int64_t rows_per_batch, float* dx,
                         float* dmult, float* dadd, int64_t ld_d, int64_t rows, int64_t C, float eps, float* drop_out,
                         float p, uint64_t seed, uint32_t site, int32_t dropout, const pfpp_planes* ret_planes,
                         const pfpp_planes* dx_planes, pfpp_stream_t stream);
int pfpp_attn_dense_train_p(const float* qkv, float* out, float* lse, const int32_t* seq_off, const int32_t* seq_len,
                            const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len, int64_t H,
                            int64_t dh, float scale, const pfpp_planes* out_planes, pfpp_stream_t stream);
int pfpp_attn_dense_bwd_p(const float* qkv, const float* out, const float* dout, const float* lse, float* dvec,
                          float* dqkv, const int32_t* seq_off, const int32_t* seq_len, const uint8_t* key_valid,
                          int64_t kv_stride, int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                          const pfpp_planes* dqkv_planes, pfpp_stream_t stream);
int pfpp_attn_blockdiag_bwd_p(const float* qkv, const float* dout, float* dqkv, int64_t n_frag, int64_t L, int64_t H,
                              int64_t dh, float scale, const pfpp_planes* dqkv_planes, pfpp_stream_t stream);

/* ---- GEMM on pre-split fp16 planes, all three products of a Linear layer ----------------------------------
 * The PFPP_GEMM_F16X3 contraction (see pfpp_gemm) with BOTH operands given as hi/lo fp16 planes, staged by LDS-DMA:
 *   forward   y  = x . W^T   (diffusers Attention to_q/to_k/to_v/to_out, FeedForward net.0.proj / net.2:
 *                             denoiser/model/modules/attention.py:46-72; TransformerEncoderLayer linears,
 *                             verifier_transformer.py:28-37):          A = x [M,K] row-major, W [N,K] row-major
 *   dX = dY . W   (autograd of the same layers in Denoiser.training_step, denoiser.py:128-145):
 *                             A = dY [M,K=out] row-major, W [K=out, N=in] k-major (w_kmajor = 1: the weight as stored)
 *   dW = dY^T . X                                             A = dY [K=rows, M=out] k-major, W = X [K=rows, N=in] k-major
 * k-major operands are read in place (no transposed copies).  K % 32 == 0, except for dW (both operands k-major),
 * where the contraction may have any length (rows beyond K are never read).  C [M, ldc] fp32 = act(alpha * A.W + bias) + residual, or with accumulate:
 * C += alpha * A.W (+ bias + residual once).  A K split (splits > 1) goes through the workspace `ws` when one is lent
 * (dense per-chunk slabs + a reduction launch, deterministic) and through fp32 atomics otherwise (accumulate only).
 * splits = 0 / variant = 0: chosen by the library.  Results of the non-accumulating form are bit-identical to
 * pfpp_gemm(PFPP_GEMM_F16X3) on the same planes.                                                               */
/* a K split's second pass, handed back instead of launched (pfpp_gemm_planes_args.defer): the slabs of `splits` chunks wait in ws
 * (and the partial column sums in csum_ws); pfpp_slab_reduce_group adds up to PFPP_SLAB_GROUP_MAX of them in ONE launch — the six weight
 * gradients of a transformer block leave one reduction behind instead of six.  splits == 0: the GEMM wrote C itself, nothing to do. */
#define PFPP_SLAB_GROUP_MAX 8
typedef struct pfpp_slab_job {
  const float* ws; float* C; const float* csum_ws; float* csum;
  int32_t M, N; int64_t ldc; int32_t splits, accumulate;
} pfpp_slab_job;
/* C (+)= sum over the chunks, chunk order (bit-identical to the reduction pfpp_gemm_planes launches itself); jobs with splits == 0 skipped */
int pfpp_slab_reduce_group(const pfpp_slab_job* jobs, int32_t n_jobs, pfpp_stream_t stream);

typedef struct pfpp_gemm_planes_args {
  const void* a_hi; const void* a_lo;   /* fp16 planes of A */
  const void* w_hi; const void* w_lo;   /* fp16 planes of W */
  float* C;
  const float* bias;                    /* [N] or NULL */
  const float* residual;                /* [M, ldr] or NULL */
  int64_t M, N, K;                      /* output rows / columns, contraction length */
  int64_t lda, ldw, ldc, ldr;           /* plane leading dimensions in halfs, C / residual in floats */
  int32_t a_kmajor, w_kmajor;
  int32_t act;                          /* PFPP_ACT_NONE / RELU / SILU / GELU */
  int32_t accumulate;
  int32_t splits, variant;
  float alpha;
  int32_t single_pass;                  /* 1: PFPP_GEMM_F16 arithmetic (hi planes only); forward layout only */
  float* ws; int64_t ws_bytes;          /* optional K-split workspace (>= splits * M * N floats): chunks write dense slabs, a second
                                           launch adds them in chunk order (deterministic); without it a K split uses atomics   */
  float* colsum; float colsum_alpha;    /* dW form only (both operands k-major): colsum[m] += colsum_alpha * sum_k A[k][m], i.e. the bias
                                           gradient sum over rows of dY (nn.Linear backward) computed from the fragments the GEMM reads
                                           anyway; with a workspace K split it needs room for splits * M more floats             */
  pfpp_slab_job* defer;                 /* NULL, or (no bias / residual / activation): a workspace K split leaves its reduction in *defer for
                                           pfpp_slab_reduce_group; ws must stay untouched until that runs                        */
} pfpp_gemm_planes_args;

int pfpp_gemm_planes(const pfpp_gemm_planes_args* args, pfpp_stream_t stream);

/* ---- the weight gradients of one transformer block in ONE launch ------------------------------------------------
 * dW_j += dY_j^T . X_j and db_j += colsum(dY_j) for up to PFPP_DW_GROUP_MAX Linear layers whose backward shares the contraction (the
 * K token rows of the step): autograd of to_q/k/v, to_out.0, ff.net.0.proj, ff.net.2 of one EncoderLayer
 * (denoiser/model/modules/attention.py:46-72, 77-90) in Denoiser.training_step (denoiser.py:128-145).  dy [K, M = out] and x [K, N = in]
 * are k-major fp16 planes read in place (leading dimensions M and N; scales folded out: gw += dY^T.X / (dy.scale * x.scale),
 * gb += colsum(dY) / dy.scale; gb may be NULL).  Every output tile runs the whole contraction in one accumulator chain (no K split,
 * no workspace, no reduction launch; deterministic); the problems' tiles are dealt to the XCDs as one concatenated list.
 * variant: 0 = the library's choice (= 2: 256 x 128 tiles, 160 workgroups for a block's six problems, unless the environment
 * names another one in PFPP_DW_GROUP_VARIANT); alternatives: 3 = 128 x 128 tiles, 6 / 7 = 128 x 64 (three / two stages).       */
#define PFPP_DW_GROUP_MAX 8
typedef struct pfpp_dw_job {
  pfpp_planes dy;      /* [K, M] */
  pfpp_planes x;       /* [K, N] */
  float* gw;           /* [M, N] fp32, accumulated in place */
  float* gb;           /* [M] fp32 or NULL */
  int64_t M, N;
} pfpp_dw_job;
int pfpp_gemm_dw_group(const pfpp_dw_job* jobs, int32_t n_jobs, int64_t K, int32_t variant, pfpp_stream_t stream);
/* name of the kernel instantiation the calling thread's last pfpp_gemm / pfpp_gemm_planes call launched through the plane
 * path ("" when that call took another kernel): lets a profiler-side tool attribute event timings to kernel names */
const char* pfpp_last_gemm_kernel(void);

/* ---- a4 + a5 + a6 fused, for a set-abstraction level without input features ---------------------------
 * PointNetSetAbstraction.forward with points = None (utils/pn2_utils.py:197-217; sa1 of PN2,
 * vqvae/model/modules/pn2.py:16): sample_and_group's centred neighbourhoods (:127-151), three
 * [1x1 conv -> BatchNorm2d (eval: folded into scale s_i / shift t_i, see pfpp_gemm) -> ReLU] and torch.max over
 * nsample (:216), in one kernel: out [F*S, C3].  idx = the ball-query result [F,S,ns]; w*_hi / w*_lo = pre-split
 * fp16 planes of the folded weights [C1, 8] (K = 3 + a zero, padded to 8), [C2, C1], [C3, C2].
 * Same arithmetic as pfpp_group_gather + 3 x pfpp_gemm(PFPP_GEMM_F16X3).  ns == 32, (C1, C2, C3) == (64, 64, 128). */
int pfpp_sa_mlp3_fused(const float* xyz, const float* new_xyz, const int32_t* idx,
                       const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                       const void* w2_hi, const void* w2_lo, const float* s0, const float* t0, const float* s1,
                       const float* t1, const float* s2, const float* t2, float* out, int64_t F, int64_t N,
                       int64_t S, int64_t ns, int64_t C1, int64_t C2, int64_t C3, pfpp_stream_t stream);

/* Same for a level WITH input features (sa2 of PN2, pn2.py:17): grouping (pn2_utils.py:127-151, rows
 * [feats[f, idx] (D) | xyz[f, idx] - new_xyz (3)]) and the FIRST TWO [1x1 conv -> BatchNorm2d(eval, folded) -> ReLU]
 * (:210-213) in one kernel: out [F*S*ns, C2] = the input of the level's third convolution (then pfpp_gemm with
 * pool = ns).  w0 planes [C1, D + 8] in the column order of pfpp_group_gather (features first), w1 planes [C2, C1].
 * ns == 64, D == C1 == C2 == 128. */
int pfpp_sa_mlp2_fused(const float* feats, const float* xyz, const float* new_xyz, const int32_t* idx,
                       const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                       const float* s0, const float* t0, const float* s1, const float* t1, float* out,
                       int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D, int64_t C1, int64_t C2,
                       pfpp_stream_t stream);
/* the same with the result (also / only) as split-f16 planes of scale 1 — the A operand of the third convolution's plane GEMM
 * (csrc/gemm_pl.hip) with no conversion pass; `out` may then be NULL */
int pfpp_sa_mlp2_fused_p(const float* feats, const float* xyz, const float* new_xyz, const int32_t* idx,
                         const void* w0_hi, const void* w0_lo, const void* w1_hi, const void* w1_lo,
                         const float* s0, const float* t0, const float* s1, const float* t1, float* out,
                         const pfpp_planes* out_planes, int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D, int64_t C1,
                         int64_t C2, pfpp_stream_t stream);
/* The same level with its FIRST convolution taken per point instead of per grouped row (it is linear): u [F*N, C1] = [feats | xyz] .
 * W1^T without bias (pfpp_gemm's fused grouping with the identity index and zero centroids; the folded shift t0 carries the bias), and
 * the value on the grouped row (s, p) is u[p] - W1_xyz . new_xyz[s].  out = the planes pfpp_sa_mlp2_fused_p writes, within fp32 rounding
 * of W1_xyz . (x - c) against W1_xyz . x - W1_xyz . c; 42 GFLOP of grouped first-layer work at F = 154 are not computed.
 * w0 planes are read for their three xyz columns only.  max_workgroups: 0 or the CU count the stream may use. */
int pfpp_sa_mlp2_table_p(const float* u, const float* new_xyz, const int32_t* idx, const void* w0_hi, const void* w0_lo,
                         const void* w1_hi, const void* w1_lo, const float* s0, const float* t0, const float* s1, const float* t1,
                         const pfpp_planes* out, int64_t F, int64_t N, int64_t S, int64_t ns, int64_t D, int64_t C1, int64_t C2,
                         int64_t max_workgroups, pfpp_stream_t stream);
/* First layer alone, for the levels whose second layer is a plane GEMM (sa3 in eval mode): out = relu(s0 * (u[idx] - W1_xyz . new_xyz) + t0)
 * as split-f16 planes [F*S*ns, C1] — what pfpp_gemm's fused grouping with the folded BatchNorm epilogue writes, as an elementwise pass over
 * the per-point table (no matrix work on the F*S*ns grouped rows).  ns == 64; (D, C1) = (256, 256) or (128, 128). */
int pfpp_sa_table_planes(const float* u, const float* new_xyz, const int32_t* idx, const void* w0_hi, const void* w0_lo,
                         const float* s0, const float* t0, const pfpp_planes* out, int64_t F, int64_t N, int64_t S, int64_t ns,
                         int64_t D, int64_t C1, pfpp_stream_t stream);

/* ---- a5 in TRAIN mode (the frozen encoder stays in .train(): train_denoiser.py:33-35, utils/pn2_utils.py:203-216 with
 * BatchNorm2d on BATCH statistics) without writing the layers' activations: one launch per layer ("stage") of the chain.
 * Stage k recomputes the chain from the level's input (grouping by ball-query index included, pn2_utils.py:127-151) through
 * layers 1..k-1 — normalised with their already finalised statistics: relu(fma(conv + bias, a_mul, a_add)), the (a_mul, a_add)
 * of pfpp_bn_finalize — and accumulates sum(y_k), sum(y_k^2) of the raw layer-k output y_k = conv + bias into `stats`
 * ([stats_copies][2][C_k] doubles, the buffer pfpp_bn_finalize reads and clears).
 *   feats == NULL (sa1, pn2.py:16): ns == 32, (C1, C2, C3) == (64, 64, 128), stages 1..3; stage 3 also writes the
 *     per-neighbourhood max and min of y_3 (out_max, out_min [F*S, C3]; then pfpp_bn_minmax_apply).
 *   feats != NULL (sa2, pn2.py:17): ns == 64, D == C1 == C2 == 128, C3 == 256, stages 1..3; stage 2 also writes the raw rows
 *     y_2 (y_out [F*S*ns, C2]: the third layer's weights do not fit in LDS next to the others); stage 3 READS y_out, applies
 *     relu(fma(y_2, a_mul[1], a_add[1])), runs the third convolution with its weight planes resident in LDS and writes the sums
 *     of y_3 and out_max / out_min [F*S, C3] (only w_hi[2] / w_lo[2] / bias[2] / a_mul[1] / a_add[1] are read).
 *   feats != NULL with D == 256 (sa3, pn2.py:18): ns == 64, widths 256 / 256 / 512.  No two of its weight matrices fit in LDS together,
 *     so every stage is ONE layer: stage 1 gathers and writes the raw rows y_1 (y_out [F*S*ns, 256]), stage 2 reads them (y_in),
 *     normalises with a_mul[0] / a_add[0] and writes y_2 (y_out), stage 3 reads y_2 (y_in, a_mul[1] / a_add[1]) and writes the sums and
 *     out_max / out_min [F*S, 512]; a workgroup keeps a 128-column slice of the layer's weight planes in LDS.
 * Weight planes as for pfpp_sa_mlp3_fused / pfpp_sa_mlp2_fused (raw conv weights, BatchNorm NOT folded).
 * max_workgroups: persistent grid size (0 = 256, one workgroup per CU; pass the CU count of a CU-masked stream). */
typedef struct pfpp_sa_train_args {
  const float* xyz;            /* [F, N, 3] */
  const float* new_xyz;        /* [F, S, 3] */
  const float* feats;          /* [F, N, D] or NULL */
  const int32_t* idx;          /* [F, S, ns] */
  const void* w_hi[3];         /* fp16 planes of the conv weights, layers 1..3 (those beyond `stage` may be NULL) */
  const void* w_lo[3];
  const float* bias[3];        /* conv biases */
  const float* a_mul[2];       /* finalised BatchNorm affine of layers 1, 2 (needed for layers < stage) */
  const float* a_add[2];
  double* stats;               /* [stats_copies][2][C_stage] */
  int64_t stats_copies;
  float* y_out;                /* feats != NULL: written by stage 2, read by stage 3 (D == 256: written by stages 1 and 2) */
  const float* y_in;           /* D == 256 only: the raw rows of the previous layer (stages 2 and 3) */
  const float* u_in;           /* optional, levels with features: U [F*N, C1] = the first conv applied PER POINT ([feats | xyz] . W1^T + b1,
                                  i.e. pfpp_gemm's fused grouping with the identity index and zero centroids).  conv1 is linear, so on a
                                  grouped row it equals U[idx] - W1_xyz . centroid: with u_in stage 1 only takes the statistics of that
                                  (no matrix work) and stage 2 gathers its input rows from U (writes y_out [F*S*ns, C2]); stage 3 is
                                  unchanged */
  float* out_max;              /* stage 3 */
  float* out_min;
  int64_t F, N, S, ns, D, C1, C2, C3;
  int32_t stage;
  int64_t max_workgroups;
  const int32_t* sched;        /* optional, ns == 64: pfpp_sa_pad_schedule(idx).  A neighbourhood with at most 32 points in range is taken as
                                  ONE half of 32 rows (its rows 32..63 are copies of row 0: 32 y_0 and 32 y_0^2 to the sums, nothing to max /
                                  min).  Of the raw second-layer rows stage 2 writes and stage 3 reads the LIVE ones only (slots below the
                                  neighbourhood's count; a copy is read as row 0), so the stages of one level must all get the schedule or
                                  none.  Table-fed stages 1-2 (u_in), stage 3 of both levels */
} pfpp_sa_train_args;
int pfpp_sa_train_stage(const pfpp_sa_train_args* args, pfpp_stream_t stream);

/* Padding schedule of a 64-neighbour level for pfpp_sa_train_args.sched.  query_ball_point (utils/pn2_utils.py:103-123) sorts the in-range
 * point indices ascending and fills the remaining nsample slots with the first one: slots cnt .. 63 repeat slot 0.  cnt = 1 + the highest
 * slot that differs from slot 0 is found slot by slot, so the schedule is exact for any index list.
 * sched [3 G + 1] int32: [0, G) the neighbourhoods with cnt > 32 (two live halves; ascending), then those with one; [G] the number of the
 * former; [G + 1, 2 G + 1) cnt by neighbourhood; [2 G + 1, 3 G + 1) cnt in schedule order.  idx [G, 64].  Two small launches per level. */
int pfpp_sa_pad_schedule(const int32_t* idx, int64_t G, int64_t ns, int32_t* sched, pfpp_stream_t stream);


/* ---- a7/a8: vector quantisation + scatter ----------------------------------
 * VectorQuantizer.forward, vqvae/model/modules/quantizer.py:26-71 as used by
 * VQVAE.encode (denoiser/model/modules/encoder.py:20-38): for every
 * `dim`-wide sub-vector z: d_j = (|z|^2 + |e_j|^2) - 2*(z.e_j), first argmin,
 * out = z + (e_j - z) (straight-through value, quantizer.py:63).  Results are
 * scattered straight into the zero-initialised padded tensor at row
 * slot[f] (denoiser.py:72-76): z_e [F, rows_per_frag, dim] ->
 * z_q [n_slots, rows_per_frag, dim]; codes [F, rows_per_frag] (may be NULL).
 * dim == 16, n_codes <= 1024.                                               */
int pfpp_vq_encode(const float* z_e, const float* codebook, const int32_t* slot,
                   float* z_q, int32_t* codes, int64_t F, int64_t rows_per_frag,
                   int64_t dim, int64_t n_codes, pfpp_stream_t stream);

/* scatter rows: out[slot[f], :] = in[f, :]  (xyz scatter, denoiser.py:76)   */
int pfpp_scatter_rows(const float* in, const int32_t* slot, float* out,
                      int64_t F, int64_t row_elems, pfpp_stream_t stream);

/* ---- a9: token features ---------------------------------------------------
 * DenoiserTransformer._gen_cond, denoiser_transformer.py:117-135 with
 * EmbedderNerf.embed utils/model_utils.py:68-69 (include_input, 10 log-spaced
 * frequencies 2^0..2^9, [sin, cos] per frequency):
 *   shape_feat[(bp,l), :] = [latent(64) | PE(xyz)(63) | PE(scale)(21) | 0-pad]
 *   pose_feat[bp, :]      = [PE(x)(147) | 0-pad]
 *   latent [n, L, 64], xyz [n, L, 3], scale [n], x [n, 7];  ld = 148.       */
int pfpp_token_features(const float* latent, const float* xyz, const float* scale,
                        const float* x, float* shape_feat, float* pose_feat,
                        int64_t n, int64_t L, pfpp_stream_t stream);

/* token assembly, denoiser_transformer.py:150-156,173-185 + PositionalEncoding
 * utils/model_utils.py:18-21:
 *   tok[(b,p,l), :] = shape_emb[(b,p,l), :] + x_emb[(b,p), :]
 *                     + ref_emb[ref[b,p] ? 1 : 0, :] + pe[p, :]              */
int pfpp_token_combine(const float* shape_emb, const float* x_emb,
                       const float* ref_emb, const uint8_t* ref_part,
                       const float* pe, float* tok, int64_t B, int64_t P,
                       int64_t L, int64_t C, pfpp_stream_t stream);
/* pfpp_token_features / pfpp_token_combine_list / pfpp_token_combine_bwd for a compacted fragment list that is still stored at its
 * padded slots: listed fragment f is read from row slot[f] of latent [n_slots, L, 64], xyz, scale, x and ref_part [n_slots] — the
 * valid-fragment gather of DenoiserTransformer.forward's inputs (denoiser.py:66-77) without the five gathered copies */
int pfpp_token_features_slots(const float* latent, const float* xyz, const float* scale, const float* x,
                              const int32_t* slot, float* shape_feat, float* pose_feat, int64_t n, int64_t L,
                              pfpp_stream_t stream);
int pfpp_token_combine_slots(const float* shape_emb, const float* x_emb, const float* ref_emb,
                             const uint8_t* ref_part, const float* pe, const int32_t* frag_pos, const int32_t* slot,
                             float* tok, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream);
int pfpp_token_combine_bwd_slots(const float* dtok, const uint8_t* ref_part, const int32_t* slot, float* dx_emb,
                                 float* dref_emb, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream);
/* same for a compacted fragment list (padded slots dropped): n fragments, frag_pos[f] = p (row of pe) */
int pfpp_token_combine_list(const float* shape_emb, const float* x_emb,
                            const float* ref_emb, const uint8_t* ref_part,
                            const float* pe, const int32_t* frag_pos, float* tok,
                            int64_t n, int64_t L, int64_t C, pfpp_stream_t stream);

/* ---- a11: AdaLN ---------------------------------------------------------------
 * MyAdaLayerNorm, denoiser/model/modules/attention.py:21-25.
 * pfpp_silu_embed: out[i, b, :] = silu(tables[i][t[b], :]) for the n_tab
 * embedding tables (one per norm); tables [n_tab, n_emb, C] is a packed copy.
 * pfpp_layernorm: y = LN(x) (eps, biased variance, no affine) then
 *   mod != NULL : y*(1+mod[b, 0:C]) + mod[b, C:2C]   (b = row / rows_per_batch)
 *   gamma != NULL : y*gamma + beta                   (norm3 / verifier norms)
 * x may alias y.                                                              */
int pfpp_silu_embed(const float* tables, const int64_t* t, float* out,
                    int64_t n_tab, int64_t n_emb, int64_t B, int64_t C,
                    pfpp_stream_t stream);
int pfpp_layernorm(const float* x, float* y, const float* mod, int64_t ld_mod,
                   const float* gamma, const float* beta, int64_t rows,
                   int64_t C, int64_t rows_per_batch, float eps,
                   pfpp_stream_t stream);
/* batch of a row given by an explicit per-group map instead of row / rows_per_batch:
 * b = group_batch[row / group_rows]  (compacted fragment lists: group = fragment, group_rows = L) */
int pfpp_layernorm_grouped(const float* x, float* y, const float* mod, int64_t ld_mod,
                           const int32_t* group_batch, int64_t group_rows, int64_t rows,
                           int64_t C, float eps, pfpp_stream_t stream);
/* same, but the result is written as split-f16 planes (hi, lo) for the PFPP_GEMM_F16X3 path */
int pfpp_layernorm_split(const float* x, void* y_hi, void* y_lo, const float* mod, int64_t ld_mod,
                         const float* gamma, const float* beta, int64_t rows,
                         int64_t C, int64_t rows_per_batch, float eps,
                         pfpp_stream_t stream);
/* pfpp_layernorm_grouped with the normalised rows written as split-f16 planes (input of the next GEMM) */
int pfpp_layernorm_grouped_split(const float* x, void* y_hi, void* y_lo, const float* mod, int64_t ld_mod,
                                 const int32_t* group_batch, int64_t group_rows, int64_t rows, int64_t C,
                                 float eps, pfpp_stream_t stream);

/* ---- a10/a12: block-diagonal self-attention --------------------------------
 * EncoderLayer self-attn (attention.py:77-80) with the block-diagonal mask of
 * DenoiserTransformer._gen_mask (denoiser_transformer.py:158-162): every
 * fragment's L tokens attend only to each other, so the [B,T,T] mask is never
 * built.  qkv [n_frag*L, 3*H*dh] (q | k | v), out [n_frag*L, H*dh];
 * softmax(q.k^T * scale).  L <= 32, dh == 64.                               */
int pfpp_attn_blockdiag(const float* qkv, float* out, int64_t n_frag, int64_t L,
                        int64_t H, int64_t dh, float scale, pfpp_stream_t stream);
/* out_hi/out_lo != NULL: additionally/instead write split-f16 planes (out may then be NULL) */
int pfpp_attn_blockdiag_split(const float* qkv, void* out_hi, void* out_lo, int64_t n_frag, int64_t L,
                              int64_t H, int64_t dh, float scale, pfpp_stream_t stream);

/* ---- a13/a18: masked row softmax for the dense attentions -------------------
 * S [rows_total, ld] in place: p = softmax(S[r, 0:T] * scale) over the keys j
 * with key_valid[batch(r), j] != 0 (batch(r) = r / rows_per_batch); masked
 * keys and the pad columns [T, ld) get exactly 0.  key_valid [n_batch, T]
 * (uint8): gen_mask of denoiser_transformer.py:163-164 / src_key_padding_mask
 * of verifier_transformer.py:62.                                             */
int pfpp_softmax_rows(float* S, const uint8_t* key_valid, int64_t rows_total,
                      int64_t rows_per_batch, int64_t T, int64_t ld, float scale,
                      pfpp_stream_t stream);

/* ---- a13/a18: fused dense masked attention ------------------------------------------------------
 * softmax(Q K^T * scale + key mask) V per (sequence, head) straight from a packed projection
 * qkv [rows, 3*H*dh] (q | k | v): EncoderLayer global attention (attention.py:82-85 with gen_mask of
 * denoiser_transformer.py:163-164) and the verifier's self-attention (verifier_transformer.py:62,
 * src_key_padding_mask).  Scores stay in registers (online softmax); exact fp32 MFMA products.
 * Sequence s occupies rows [seq_off[s], seq_off[s] + seq_len[s]); key_valid (may be NULL) is
 * [n_seq, kv_stride] uint8 with 0 = key masked out.  out [rows, H*dh].  dh in {32, 64}.         */
int pfpp_attn_dense(const float* qkv, float* out, const int32_t* seq_off, const int32_t* seq_len,
                    const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                    int64_t H, int64_t dh, float scale, pfpp_stream_t stream);
int pfpp_attn_dense_split(const float* qkv, void* out_hi, void* out_lo, const int32_t* seq_off,
                          const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride,
                          int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                          pfpp_stream_t stream);

/* ---- a15: mean pool over the L tokens of a fragment -----------------------
 * denoiser_transformer.py:139-142.  x [n*L, C] -> out [n, C]                 */
int pfpp_mean_pool(const float* x, float* out, int64_t n, int64_t L, int64_t C,
                   pfpp_stream_t stream);

/* ---- a16: DDPM ancestral step + reference-part re-pin ----------------------
 * diffusers-0.21.4 DDPMScheduler.step as configured by PiecewiseScheduler
 * (denoiser/model/modules/custom_diffusers.py:60-69) followed by
 * `x[ref_part] = reference[ref_part]` (denoiser.py:184-185, auto_aggl.py:149-150)
 *   x0 = (x - c_eps*eps)/c_div;  out = c_x0*x0 + c_x*x (+ c_noise*noise)
 * coefficients are computed on the host in fp32 in the scheduler's op order;
 * noise may be NULL (t == 0).  All tensors [n, 7]; ref_part [n] uint8 or NULL */
int pfpp_ddpm_step(const float* x, const float* eps, const float* noise,
                   const uint8_t* ref_part, const float* reference, float* out,
                   int64_t n, float c_eps, float c_div, float c_x0, float c_x,
                   float c_noise, pfpp_stream_t stream);

/* add_noise (denoiser.py:92): out = sa[b]*x0 + sb[b]*noise, per-puzzle scalars */
int pfpp_add_noise(const float* x0, const float* noise, const float* sqrt_ab,
                   const float* sqrt_1mab, float* out, int64_t B,
                   int64_t per_batch, pfpp_stream_t stream);

/* ---- a18: verifier token embedding ------------------------------------------
 * verifier_transformer.py:52-56: tok = feat_emb[(b,e), :] + [pe[i0] | pe[i1]]
 * feat_emb [n, C] (output of the 7->C GEMM), edge_idx [n, 2] int64,
 * pe [max_len, C/2].                                                         */
int pfpp_verifier_embed(const float* feat_emb, const int64_t* edge_idx,
                        const float* pe, float* tok, int64_t n, int64_t C,
                        int64_t max_len, pfpp_stream_t stream);

/* ---- a19: pose composition ---------------------------------------------------
 * utils/node_merge_utils.py:275-306 get_param / :246-272
 * extract_final_pred_trans_rots: M = [R(q[pivot[i]]) | t[pivot[i]]], optionally
 * M <- M @ init_pose[i] (has_init[i] != 0), then (t, matrix_to_quaternion(R)).
 * pose [P,7], pivot [n] int32, init_pose [n,16] row-major 4x4, out [n,7].
 * quaternion_to_matrix / matrix_to_quaternion follow pytorch3d.  */
int pfpp_pose_compose(const float* pose, const int32_t* pivot,
                      const float* init_pose, const uint8_t* has_init,
                      float* out, int64_t n, pfpp_stream_t stream);

/* ---- 8f-1: verifier edge features -----------------------------------------------------------------
 * pfpp_pose_apply_points: body of get_final_pose_pts_dynamic (utils/node_merge_utils.py:16-41): a flat
 * list of points, point i is moved by pose[pose_idx[i]] = (t, q): quaternion_apply (normalise = 0 there)
 * then + t.
 * pfpp_edge_histogram: get_distance_for_matching_pts (:62-89) + _make_cd_to_bins (auto_aggl.py:385-389):
 * edge e owns the matched pairs [edge_off[e], edge_off[e+1]) of (idx_a, idx_b) (indices into pts);
 * d_i = min_j |a_i-b_j|^2 + min_j |b_i-a_j|^2 is counted into the 6 bins
 * [0,1e-3) [1e-3,5e-3) [5e-3,1e-2) [1e-2,5e-2) [5e-2,1e-1) [1e-1,100) -> hist [n_edges, 6] int32.  */
int pfpp_pose_apply_points(const float* pts, const int32_t* pose_idx, const float* pose, float* out,
                           int64_t n, int normalise, pfpp_stream_t stream);
int pfpp_edge_histogram(const float* pts, const int32_t* idx_a, const int32_t* idx_b,
                        const int32_t* edge_off, int32_t* hist, int64_t n_edges, int64_t max_m,
                        pfpp_stream_t stream);

/* ---- 8f-4: GPU-side augmentation of GeometryLatentDataset.__getitem__ (denoiser/dataset/dataset.py:165-215) ---
 * For a batch of stored puzzles part_pcs_gt [B,P,N,3] (assembled frame, padded): rotate the whole assembly by
 * R(q_global)^T, recentre on the reference part, then per part recentre + rotate by R(q_part)^T and divide by
 * the max-abs coordinate.  q_global [B,4] / q_part [B,P,4] are the quaternions the dataset stores (pose_gt_r /
 * part_rots, scalar first).  Outputs: part_pcs [B,P,N,3], part_trans [B,P,3], part_scale [B,P], init_pose_t [B,3];
 * padded slots (p >= num_parts[b]) get zeros and scale 1.  float64 means/rotations like the numpy original.
 * workspace: pfpp_fragment_prepare_workspace(B, P) bytes.  N <= 2048.                                       */
int64_t pfpp_fragment_prepare_workspace(int64_t B, int64_t P);
int pfpp_fragment_prepare(const float* part_pcs_gt, const int32_t* num_parts, const int32_t* ref_idx,
                          const float* q_global, const float* q_part, float* part_pcs, float* part_trans,
                          float* part_scale, float* init_pose_t, int64_t B, int64_t P, int64_t N,
                          void* workspace, pfpp_stream_t stream);

/* ---- 8f-2: merge step of auto_aggl (utils/node_merge_utils.py:159-222) -------------------------------------
 * pfpp_estimate_normals: pytorch3d.ops.estimate_pointcloud_normals(neighborhood_size=K) per part:
 * pts [P, N, 3] -> normals [P, N, 3] (K nearest neighbours incl. the point, covariance about the
 * neighbourhood mean, eigenvector of the smallest eigenvalue, majority-side sign).  K in {10, 20, 32}.
 * pfpp_merge_keep_mask: keep[i,k] = 0 iff for some j != i  d[i,j,k] + d[j,i,k] < threshold and
 * normals[i,k].normals[j,k] < 0;  d [P,P,N] = nearest-neighbour distances part i -> part j (pfpp_nn_dist).
 * pfpp_fps_start: pfpp_fps with an explicit first index per fragment (torch_cluster.fps random_start,
 * node_merge_utils.py:219); N up to 32768.                                                              */
int pfpp_estimate_normals(const float* pts, float* normals, int64_t P, int64_t N, int64_t K,
                          pfpp_stream_t stream);
int pfpp_merge_keep_mask(const float* d, const float* normals, uint8_t* keep, int64_t P, int64_t N,
                         float threshold, pfpp_stream_t stream);
int pfpp_fps_start(const float* xyz, int32_t* idx, float* new_xyz, int64_t F, int64_t N, int64_t S,
                   const int32_t* start, pfpp_stream_t stream);

/* ---- 8f-3: evaluation metrics (denoiser/evaluation/evaluator.py, transform.py) ---------------------------
 * pfpp_nn_dist: out[b, i] = min_j |src[b,i] - dst[b,j]|^2 — the KNN-1 term of chamferdist.ChamferDistance
 * (squared L2, knn_points); calc_part_acc (evaluator.py:88-121) and calc_shape_cd (:124-153) call it in
 * both directions.  src [batch, n, 3], dst [batch, m, 3], out [batch, n].
 * pfpp_quat_to_euler_xyz: transform.quaternion_to_euler (transform.py:70-86) = pytorch3d
 * quaternion_to_matrix + matrix_to_euler_angles("XYZ"); quat [n,4] (w first) -> euler [n,3].             */
int pfpp_nn_dist(const float* src, const float* dst, float* out, int64_t batch, int64_t n, int64_t m,
                 pfpp_stream_t stream);
int pfpp_quat_to_euler_xyz(const float* quat, float* euler, int64_t n, int to_degree,
                           pfpp_stream_t stream);

/* =====================================================================================================
 * a17: training — backward of the DenoiserTransformer, loss and optimizer
 * (Denoiser.forward/_loss/training_step/configure_optimizers, denoiser/model/denoiser.py:80-145,230-241).
 * The encoder is frozen in the reference (train_denoiser.py:33-35): gradients stop at the tokens.
 * ===================================================================================================== */

/* ---- backward GEMMs (csrc/gemm_grad.hip) ----------------------------------------------------------
 *   C[M,N] (+)= alpha * sum_k A(m,k) * W(n,k)
 *   a_kmajor = 0: A(m,k) = A[m*lda + k]     a_kmajor = 1: A(m,k) = A[k*lda + m]
 *   w_kmajor = 0: W(n,k) = W[n*ldw + k]     w_kmajor = 1: W(n,k) = W[k*ldw + n]
 * dX = dY . W        : A = dY, W = the [out,in] weight with w_kmajor = 1
 * dW = dY^T . X      : A = dY (a_kmajor = 1), W = X (w_kmajor = 1), K = rows
 * split-f16 arithmetic (PFPP_GEMM_F16X3); a_scale / w_scale are powers of two applied before the
 * f16 split (gradients are tiny) and divided out of the result.  split_k = 0 picks the number of K
 * chunks; more than one chunk needs accumulate = 1 (fp32 atomics into a zero-initialised C).     */
typedef struct pfpp_gemm_grad_args {
  const float* A; const float* W; float* C;
  int64_t M, N, K;
  int64_t lda, ldw, ldc;
  int32_t a_kmajor, w_kmajor;
  int32_t accumulate;
  int32_t split_k;
  int32_t batch;
  int64_t sA, sW, sC;
  float a_scale, w_scale, alpha;
  float* colsum;        /* NULL, or (weight-gradient form: a_kmajor = 1, batch = 1) colsum[m] += sum_k A[m, k] — the bias gradient of
                           the same torch.nn.Linear (unscaled dY), accumulated with atomics from the A tiles as they are staged */
} pfpp_gemm_grad_args;
int pfpp_gemm_grad(const pfpp_gemm_grad_args* args, pfpp_stream_t stream);
/* Up to 8 independent weight-gradient problems (both operands k-major, batch 1) in ONE launch: the six dW of a transformer
 * layer (attention.py:60-90: to_q/k/v, to_out, ff.net.0.proj, ff.net.2 of the self- and global-attention blocks) each have
 * few output tiles and a contraction over all tokens; launched together they fill the chip without cutting K into many
 * atomically-accumulated chunks.  Replaces what autograd does one Linear at a time (torch.nn.Linear backward,
 * denoiser_transformer.py:79-101 layers).  split_k of each problem: 0 = chosen for the group. */
int pfpp_gemm_grad_group(const pfpp_gemm_grad_args* args, int count, pfpp_stream_t stream);

/* out[z, c] (+)= sum_r x[z, r, c]  (bias gradients).  x rows of stride ld, batches of stride sx / so */
int pfpp_colsum(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld,
                int64_t batch, int64_t sx, int64_t so, int accumulate, pfpp_stream_t stream);

/* ---- counter-based dropout --------------------------------------------------------------------------
 * keep(i) = rng(seed, site, i) >= p * 2^32 ; y = x * keep / (1 - p).  The same (seed, site) gives the
 * same mask in forward and backward, so masks are never stored.  Sites of the denoiser: token dropout
 * (PositionalEncoding, utils/model_utils.py:18-21, p = 0.1), the dropout after each attention
 * out-projection and inside each FeedForward (diffusers Attention.to_out[1] / FeedForward.net[1],
 * attention.py:46-72 with dropout_rate of config/denoiser/model.yaml).
 * pfpp_dropout: out = (res ? res : 0) + x*keep/(1-p)   (x may alias out)
 * pfpp_dropout_mask: the keep mask itself as uint8 (tests, oracle)                                  */
int pfpp_dropout(const float* x, const float* res, float* out, int64_t n, float p,
                 uint64_t seed, uint32_t site, pfpp_stream_t stream);
int pfpp_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint32_t site,
                      pfpp_stream_t stream);

/* ---- GEGLU (diffusers FeedForward "geglu", attention.py:67-72) with its dropout, training form ------
 * z [rows, 2*inner] = proj output (value columns [0,inner), gate columns [inner, 2*inner))
 * fwd: u[r,c] = drop(z[r,c] * gelu_erf(z[r,inner+c]))
 * bwd: dz[r,c] = du' * gelu(g) ; dz[r,inner+c] = du' * v * gelu'(g) with du' = du*keep/(1-p)        */
int pfpp_geglu(const float* z, float* u, int64_t rows, int64_t inner, float p, uint64_t seed,
               uint32_t site, pfpp_stream_t stream);
int pfpp_geglu_bwd(const float* z, const float* du, float* dz, int64_t rows, int64_t inner, float p,
                   uint64_t seed, uint32_t site, pfpp_stream_t stream);

/* elementwise activation and its backward on a flat buffer (output heads: SiLU)                     */
int pfpp_act(const float* pre, float* out, int64_t n, int act, pfpp_stream_t stream);
int pfpp_act_bwd(const float* pre, const float* dy, float* dx, int64_t n, int act, pfpp_stream_t stream);

/* ---- LayerNorm backward (MyAdaLayerNorm attention.py:21-25, norm3) ----------------------------------
 * y = xhat * mult + add with  mult = 1 + mod[b, 0:C], add = mod[b, C:2C]  (mod != NULL, AdaLN)
 *                       or    mult = gamma, add = beta                    (gamma != NULL)
 *   dx[r, :]      += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * mult
 *   dmult[b, c]   += sum_r dy * xhat       dadd[b, c] += sum_r dy      (b = 0 for the affine form)
 * batch of a row: group_batch[row / group_rows] if group_batch else row / rows_per_batch; group_rows
 * rows are handled per workgroup (rows_per_batch % group_rows == 0).  dmult/dadd rows have stride ld_d.
 * C in {256, 512}.                                                                                   */
int pfpp_layernorm_bwd(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                       const float* gamma, const int32_t* group_batch, int64_t group_rows,
                       int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                       int64_t rows, int64_t C, float eps, pfpp_stream_t stream);

/* The same with the dropout that follows each LayerNorm in the backward chain fused in (EncoderLayer's
 * dropout after the attention out-projections attention.py:46-52, PositionalEncoding's token dropout
 * model_utils.py:18-21): drop_out[r, c] = keep(seed, site, r*C + c) ? dx_new[r, c] / (1 - p) : 0 — the mask of
 * pfpp_dropout over a [rows, C] tensor, so forward and backward sites regenerate each other's masks.      */
int pfpp_layernorm_bwd_dropout(const float* x, const float* dy, const float* mod, int64_t ld_mod,
                               const float* gamma, const int32_t* group_batch, int64_t group_rows,
                               int64_t rows_per_batch, float* dx, float* dmult, float* dadd, int64_t ld_d,
                               int64_t rows, int64_t C, float eps, float* drop_out, float p, uint64_t seed,
                               uint32_t site, pfpp_stream_t stream);

/* Forward counterpart: h_out = (res or 0) + dropout(y; p, seed, site), n_out = LayerNorm(h_out) with the
 * (mod | gamma, beta | none) forms and the row -> batch mapping of pfpp_layernorm / pfpp_layernorm_grouped
 * (group_batch != NULL: batch = group_batch[row / group_rows], else row / rows_per_batch).  h_out may alias
 * y or res.  h_out has the bits of pfpp_dropout, n_out those of pfpp_layernorm up to the contraction of the
 * affine step (an ulp).  C in {256, 512}.                                                                 */
int pfpp_dropout_layernorm(const float* y, const float* res, float* h_out, float* n_out, const float* mod,
                           int64_t ld_mod, const float* gamma, const float* beta, const int32_t* group_batch,
                           int64_t group_rows, int64_t rows_per_batch, int64_t rows, int64_t C, float eps,
                           float p, uint64_t seed, uint32_t site, pfpp_stream_t stream);

/* ---- attention backward ---------------------------------------------------------------------------
 * dqkv [rows, 3*H*dh] receives (dq | dk | dv) of softmax(q.k^T * scale) v.
 * pfpp_attn_dense_train: pfpp_attn_dense that also writes lse[row, h] = log sum_j exp(s_ij) (needed by
 * the backward).  pfpp_attn_dense_bwd: two passes without atomics — dq per query block, dk/dv per key
 * block, both recomputing the probabilities from q, k and lse.                                       */
int pfpp_attn_blockdiag_bwd(const float* qkv, const float* dout, float* dqkv, int64_t n_frag,
                            int64_t L, int64_t H, int64_t dh, float scale, pfpp_stream_t stream);
int pfpp_attn_dense_train(const float* qkv, float* out, float* lse, const int32_t* seq_off,
                          const int32_t* seq_len, const uint8_t* key_valid, int64_t kv_stride,
                          int64_t n_seq, int64_t max_len, int64_t H, int64_t dh, float scale,
                          pfpp_stream_t stream);
int pfpp_attn_dense_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                        float* dvec /* workspace [rows, H] */, float* dqkv, const int32_t* seq_off, const int32_t* seq_len,
                        const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                        int64_t H, int64_t dh, float scale, pfpp_stream_t stream);
/* the same in separately launchable parts (bit 0: D = rowsum(dout . out) -> dvec, bit 1: dq, bit 2: dk/dv): once D is
 * there dq and dk/dv are independent (disjoint columns of dqkv) and may be issued on two streams.               */
int pfpp_attn_dense_bwd_parts(const float* qkv, const float* out, const float* dout, const float* lse,
                              float* dvec, float* dqkv, const int32_t* seq_off, const int32_t* seq_len,
                              const uint8_t* key_valid, int64_t kv_stride, int64_t n_seq, int64_t max_len,
                              int64_t H, int64_t dh, float scale, int parts, pfpp_stream_t stream);

/* ---- small backward pieces ----------------------------------------------------------------------------
 * mean_pool_bwd:   dx[(f,l), :] = dpooled[f, :] / L                         (denoiser_transformer.py:139-142)
 * token_combine_bwd: dx_emb[f, :] = sum_l dtok[(f,l), :]; dref_emb[ref[f], :] += dx_emb[f, :]
 *                  (the shape-embedding gradient is dtok itself; pe is a buffer)
 * silu_embed_bwd:  dtables[i][t[b], :] += dse[i, b, :] * silu'(tables[i][t[b], :]); puzzles with equal t[b] are summed in index
 *                  order without atomics (deterministic: data-parallel ranks scattering the same gathered list stay bit-equal) */
int pfpp_mean_pool_bwd(const float* dpooled, float* dx, int64_t n, int64_t L, int64_t C,
                       pfpp_stream_t stream);
int pfpp_token_combine_bwd(const float* dtok, const uint8_t* ref_part, float* dx_emb, float* dref_emb,
                           int64_t n, int64_t L, int64_t C, pfpp_stream_t stream);
int pfpp_silu_embed_bwd(const float* tables, const int64_t* t, const float* dse, float* dtables,
                        int64_t n_tab, int64_t n_emb, int64_t B, int64_t C, pfpp_stream_t stream);
/* the same, and the rows t[.] are marked in `active` (uint32 [ceil(n_emb / 32)], bit r = row r has received a gradient since the bitmap
 * was cleared) for pfpp_adamw_rows_active; n_emb <= 4096 */
int pfpp_silu_embed_bwd_mark(const float* tables, const int64_t* t, const float* dse, float* dtables,
                             int64_t n_tab, int64_t n_emb, int64_t B, int64_t C, uint32_t* active, pfpp_stream_t stream);

/* ---- backward of the AdaLN modulation linears (MyAdaLayerNorm.forward, attention.py:21-25: mods_j = Linear_j(silu(emb_j(t))), one per
 * norm, 2 * num_layers of them) in two launches per 32 puzzles, plain fp32, fixed summation order (csrc/ada_bwd.hip):
 *   g_w[j][n, k] += sum_b dmods[j][b, n] se[j][b, k]     g_b[j][n] += sum_b dmods[j][b, n]     dse[j][b, k] = sum_n dmods[j][b, n] w[j][n, k]
 * dmods [n_ada, B, N2], se / dse [n_ada, B, C], w / g_w [n_ada, N2, C], g_b [n_ada, N2]; scratch: pfpp_ada_linear_bwd_scratch_floats()
 * floats (16-byte aligned).  Replaces a column sum and two tiled gradient GEMMs (the contraction of the weight gradient is only B deep). */
/* ---- backward of the token embedding in one launch (DenoiserTransformer._gen_cond / _add_ref_part_emb / forward,
 * denoiser_transformer.py:117-135, 150-156, 173-185; csrc/embed_train.hip).  The forward leaves the EXTENDED feature rows
 *   F[m] = [shape features (148) | pose features of the token's fragment (147) | [ref_part = 0] | [ref_part = 1] | 1 | 0 ...]  (320 columns)
 * transposed as split-f16 planes ft_hi / ft_lo [320, Mp] (Mp = pfpp_token_features_t_cols(n, L): n L rounded up to 16, zero padded;
 * arguments as pfpp_token_features_slots plus ref_part [slots]); pfpp_token_embed_bwd contracts dtok [n L, C] with them over the tokens:
 *   g_shape_w [C, 148] += dtok^T sf   g_param_w [C, 147] += (sum_l dtok)^T pf   g_shape_b, g_param_b [C] += sum_m dtok[m]
 *   g_ref_emb [2, C]: row r += the sum of dtok over the tokens of fragments with ref_part = r
 * split-f16 products of (g_scale * dtok) (g_scale a power of two, divided out), fixed summation order, no atomics. */
/* training forward: the operand of pfpp_embed_tokens_small built from the fp32 parameters of the step — [W_shape (148) | W_param (147) | 0]
 * [C, 320] as fragment-blocked split-f16 planes fhi / flo (C * 320 halfs each, scale 1) and bias [C] = b_shape + b_param */
int pfpp_embed_pack_weights(const float* w_shape, const float* w_param, const float* b_shape, const float* b_param, void* fhi, void* flo,
                            float* bias, int64_t C, pfpp_stream_t stream);
int64_t pfpp_token_features_t_cols(int64_t n, int64_t L);
int pfpp_token_features_t(const float* latent, const float* xyz, const float* scale, const float* x, const int32_t* slot,
                          const uint8_t* ref_part, void* ft_hi, void* ft_lo, int64_t n, int64_t L, pfpp_stream_t stream);
int pfpp_token_embed_bwd(const float* dtok, const void* ft_hi, const void* ft_lo, float* g_shape_w, float* g_shape_b, float* g_param_w,
                         float* g_param_b, float* g_ref_emb, int64_t n, int64_t L, int64_t C, float g_scale, pfpp_stream_t stream);

int64_t pfpp_ada_linear_bwd_scratch_floats(int64_t n_ada, int64_t N2);
int pfpp_ada_linear_bwd(const float* dmods, const float* se, const float* w, float* g_w, float* g_b, float* dse, float* scratch,
                        int64_t n_ada, int64_t B, int64_t C, int64_t N2, pfpp_stream_t stream);

/* ---- loss (Denoiser._loss, denoiser.py:118-126) ---------------------------------------------------------
 * loss = mean over the selected rows (valid, non-reference fragments) x 7 of (pred - target)^2;
 * dpred = grad_out * 2 (pred - target) / (7 * n_sel) on selected rows, 0 elsewhere.  loss [1].       */
int pfpp_mse_loss(const float* pred, const float* target, const uint8_t* sel, float* loss,
                  float* dpred, int64_t n, int64_t width, float grad_out, pfpp_stream_t stream);
/* the same with the selection computed in the kernel: row r counts iff valid[r] != 0 and ref[r] == 0 (part_valids & ~ref_part,
 * denoiser.py:118-121, never materialised); amax (optional, needs dpred): receives max |dpred| of the call */
int pfpp_mse_loss_masked(const float* pred, const float* target, const float* valid, const uint8_t* ref, float* loss,
                         float* dpred, float* amax, int64_t n, int64_t width, float grad_out, pfpp_stream_t stream);

/* ---- AdamW (configure_optimizers, denoiser.py:230-241; torch.optim.AdamW semantics) ---------------------
 * p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g^2;
 * p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)      with bc1 = 1-b1^t, bc2 = 1-b2^t, g *= g_scale.
 * hi/lo (may be NULL): refreshed split-f16 planes of p for the forward GEMMs.                        */
int pfpp_adamw(float* p, const float* g, float* m, float* v, void* hi, void* lo, int64_t n,
               float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
               float bc2, float g_scale, pfpp_stream_t stream);
/* the same, and with zero_grad != 0 the gradient buffer is cleared in the same pass (optimizer.step() + optimizer.zero_grad(),
 * the pair a training loop issues back to back) */
int pfpp_adamw_zero(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                    float bc2, float g_scale, int zero_grad, pfpp_stream_t stream);
/* the same with the overflow guard of a loss-scaled optimizer (what torch.cuda.amp.GradScaler.step does around
 * configure_optimizers' AdamW, denoiser.py:230-241, when the backward runs on fp16 operands), entirely on the device:
 * overflow (int32[2], device memory): [0] = flag, [1] = count.  An element whose (scaled) gradient is inf / NaN is left untouched
 * (p, m, v, planes unchanged; its gradient is still cleared when zero_grad != 0), sets the flag and adds to the count.  The kernel
 * never READS the flag: which elements are skipped depends on their own gradient only, so the result is independent of workgroup
 * scheduling and identical on data-parallel replicas.  The caller clears overflow[] between steps and reads it when it likes. */
int pfpp_adamw_guarded(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                       float bc2, float g_scale, int zero_grad, int32_t* overflow, pfpp_stream_t stream);
/* the same update restricted by ROWS of a stack of embedding tables [n_tables, rows_per_table, C] (the 12 AdaLN timestep tables,
 * MyAdaLayerNorm.emb = nn.Embedding(num_embeds_ada_norm, dim), attention.py:18-25): a step touches only the rows of the batch's
 * timesteps t [n_t] (int64, device), every other row has an exactly zero gradient and its update (moment decay, weight decay, the
 * step of the decayed first moment) depends on nothing of this step's backward.  mode 0: every row EXCEPT t[.] — issued at the
 * start of the backward, off the iteration's exposed tail; mode 1: only the rows t[.] (each once, whatever the duplicates in t) —
 * after their gradients are final.  Both together = one pfpp_adamw_guarded over the stack, element for element.          */
int pfpp_adamw_rows(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n_tables, int64_t rows_per_table, int64_t C,
                    const int64_t* t, int64_t n_t, int mode, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float bc1, float bc2, float g_scale, int zero_grad, int32_t* overflow, pfpp_stream_t stream);
/* pfpp_adamw_guarded over the stack restricted to the rows whose bit is set in `active` (pfpp_silu_embed_bwd_mark: every row that has
 * ever received a gradient).  A row that never did has g = m = v = 0, and with weight decay as small as the reference's
 * (configure_optimizers, denoiser.py:230-237: lr 2e-4, weight_decay 1e-6 -> fl32(1 - lr wd) = 1) its AdamW update is exactly the
 * identity: of the 3,072 rows per table only the 1,000 training timesteps can ever be indexed, two thirds of the tables' 18.9 M parameters
 * are never read or written.  When fl32(1 - lr * weight_decay) != 1 every row is taken (the plain update).  The caller owns the
 * bitmap's meaning: set every bit after restoring optimizer state it does not know the history of. */
int pfpp_adamw_rows_active(float* p, float* g, float* m, float* v, void* hi, void* lo, int64_t n_tables, int64_t rows_per_table, int64_t C,
                           const uint32_t* active, float lr, float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                           float g_scale, int zero_grad, int32_t* overflow, pfpp_stream_t stream);

/* ---- train-mode BatchNorm of the (frozen, but .train()) encoder (utils/pn2_utils.py:211-214) -------------
 * The reference freezes the encoder's parameters only (train_denoiser.py:33-35); under Lightning's
 * model.train() its BatchNorm2d layers normalise with batch statistics and keep updating their buffers.
 * bn_stats: per column mean and biased variance over the rows of x [rows, C] (ld), accumulated in fp64
 * (torch's CPU kernel accumulates in double); when running_mean != NULL the running statistics are
 * updated like torch: r = (1-momentum)*r + momentum*stat, with the unbiased variance.
 * workspace: at least pfpp_bn_stats_workspace(rows, C) bytes.
 * bn_apply: y = relu(x*a + b) with a = gamma/sqrt(var+eps), b = beta - mean*a (the form ATen's CPU kernel
 * uses), optional max over groups of `pool` consecutive rows (the set-abstraction max, pn2_utils.py:216):
 * y [rows/pool, ldy].  C % 4 == 0, C <= 1024.                                                         */
/* fused form: the producing GEMM accumulates stats (see pfpp_gemm_args.stats); bn_finalize turns them into
 * mean/var (optional outputs), updates the running buffers, writes the affine a = gamma/sqrt(var+eps),
 * b = beta - mean*a for the consuming GEMM's a_mul/a_add, and clears stats for the next use.
 * bn_minmax_apply: y = relu(a*(a >= 0 ? mx : mn) + b) == max over the pool group of relu(a*x + b).    */
int pfpp_bn_finalize(double* stats, int64_t copies, int64_t rows, int64_t C, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, float* mean, float* var, float* a_mul, float* a_add,
                     pfpp_stream_t stream);
int pfpp_bn_minmax_apply(const float* mx, const float* mn, const float* a_mul, const float* a_add,
                         float* y, int64_t rows, int64_t C, pfpp_stream_t stream);
int64_t pfpp_bn_stats_workspace(int64_t rows, int64_t C);
int pfpp_bn_stats(const float* x, int64_t rows, int64_t C, int64_t ld, float* mean, float* var,
                  float* running_mean, float* running_var, float momentum, void* workspace,
                  pfpp_stream_t stream);
int pfpp_bn_apply(const float* x, int64_t rows, int64_t C, int64_t ld, const float* mean,
                  const float* var, const float* gamma, const float* beta, float eps, float* y,
                  int64_t ldy, int64_t pool, pfpp_stream_t stream);

/* ---- a17: the transformer blocks of the training step, sequenced from C ------------------------------------
 * EncoderLayer.forward (denoiser/model/modules/attention.py:74-92) for a range of the layers of
 * DenoiserTransformer.forward (denoiser_transformer.py:187-196) in train mode, and its backward (autograd of
 * Denoiser.training_step, denoiser.py:128-145): the SAME launches with the SAME arguments that the host issues one by
 * one through the entry points above (pfpp_layernorm_*_split / pfpp_dropout_layernorm_p, pfpp_gemm_planes,
 * pfpp_attn_blockdiag_split, pfpp_attn_dense_train_p, pfpp_geglu_p; backward: pfpp_gemm_planes in its dX / dW forms,
 * pfpp_geglu_bwd_p, pfpp_layernorm_bwd_p, pfpp_attn_*_bwd_p, pfpp_adamw_guarded) — enqueued from one call, so that the
 * ~240 launches of the six blocks cost the host what hipLaunchKernel costs, not a marshalled foreign call each.
 * Activations live in caller-owned arenas: layer i's saved tensors at fwd_arena + i * fwd_layer_bytes (layout private to
 * the library, pfpp_tlayers_fwd_bytes() bytes; the layer's output [M, C] fp32 sits at pfpp_tlayers_fwd_hout_offset()),
 * the backward's temporaries in bwd_arena (pfpp_tlayers_bwd_bytes()).  Weight planes are the caller's (scale 1 unless
 * stated in .scale).  Weight / bias gradients ACCUMULATE into the fp32 buffers of pfpp_tlayer_grads.                    */
typedef struct pfpp_tlayer_params {
  pfpp_planes qkv1, o1, qkv2, o2, ff1, ff2;      /* [3C,C] (to_q|to_k|to_v), [C,C], [3C,C], [C,C], [2 inner, C], [C, inner] */
  const float *bo1, *bo2, *g3, *b3, *bff1, *bff2; /* to_out.0.bias x2, norm3.weight / bias, ff.net.0.proj.bias, ff.net.2.bias */
} pfpp_tlayer_params;
typedef struct pfpp_tlayer_grads {
  float *qkv1_w, *o1_w, *o1_b, *qkv2_w, *o2_w, *o2_b, *g3, *b3, *ff1_w, *ff1_b, *ff2_w, *ff2_b;
} pfpp_tlayer_grads;
typedef struct pfpp_tlayer_adamw {              /* the layer's contiguous slice of the flat parameter buffers */
  float *p, *g, *m, *v; void *hi, *lo; int64_t n;
} pfpp_tlayer_adamw;
typedef struct pfpp_tlayers_args {
  int32_t n_layers;
  const pfpp_tlayer_params* layers;             /* [n_layers] (host memory) */
  const pfpp_tlayer_grads* grads;               /* [n_layers], backward only */
  const pfpp_tlayer_adamw* adamw;               /* [n_layers] or NULL: optimizer-in-backward on the side stream (needs side != NULL) */
  int64_t M, C, H, L, inner, Fv, B;             /* tokens (= Fv * L), width, heads, latent points, GEGLU inner width, fragments, puzzles */
  float* h_in;                                  /* [M, C] tokens; token dropout is applied IN PLACE when p_tok > 0 */
  const float* mods;                            /* [2 n_layers, B, 2C] AdaLN (scale | shift) rows */
  const int32_t* frag_b;                        /* [Fv] puzzle of every fragment */
  const int32_t *seq_off, *seq_len;             /* [n_seq] token range of every puzzle */
  int64_t n_seq, max_len;
  float att_scale, p_tok, p_lay;
  uint64_t seed;
  void* fwd_arena; int64_t fwd_layer_bytes;
  float *ws_main, *ws_side; int64_t ws_bytes;   /* K-split workspaces of pfpp_gemm_planes, one per stream */
  /* backward */
  void* bwd_arena; int64_t bwd_bytes;
  float grad_scale;                             /* power of two lifting gradient planes into the fp16 range */
  float* dh;                                    /* [M, C] running gradient of the residual stream, updated in place */
  pfpp_planes dhp;                              /* its planes (grad_scale * dh) on entry */
  pfpp_planes* dhp_out;                         /* optional: where the planes of the gradient leaving the range are (inside bwd_arena) */
  float* dmods;                                 /* [2 n_layers, B, 2C] AdaLN row gradients (accumulated) */
  float* dtok;                                  /* [M, C] gradient w.r.t. the tokens (needed when the range reaches layer 0) */
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2, opt_g_scale; int32_t opt_zero_grad; int32_t* overflow;   /* with adamw */
  /* optional (NULL: the tiled kernel everywhere): scratch of at least pfpp_tlayers_frag_bytes() for the fragment-blocked copies of a
   * layer's weights — the qkv / out / second feed-forward linears of the forward and their input gradients then run through
   * pfpp_gemm_wd.  With room for every layer of the call's range (n x pfpp_tlayers_frag_bytes()) the range's weights are blocked by ONE
   * launch at the top of the call instead of one per layer */
  void* frag_ws; int64_t frag_ws_bytes;
  /* optional (ada_se == NULL: the caller does this after the call), backward, side != NULL: the two AdaLN linears of every block
   * (MyAdaLayerNorm.linear, attention.py:21-25: mods_j = silu(table_j[t]) . W_j^T + b_j, j = 2 i, 2 i + 1) take their gradients as soon
   * as block i's backward is through — gb_j += colsum(dmods_j), gw_j += dmods_j^T . se_j, and the gradient w.r.t. the embedded timestep
   * dse_j = dmods_j . W_j — on the side stream, followed (with adamw) by the AdamW update of W_j / b_j (ada_adamw_w / _b [n_layers]:
   * the block's slices [2 i, 2 i + 2) of the stacked parameters): none of it is left for the exposed tail of the iteration */
  const float* ada_se;                          /* [2 n_layers, B, C] silu(table[t]) rows of the forward */
  float* ada_dse;                               /* [2 n_layers, B, C] out */
  const float* ada_w;                           /* [2 n_layers, 2C, C] fp32 weights (read before their update) */
  float* ada_gw; float* ada_gb;                 /* [2 n_layers, 2C, C], [2 n_layers, 2C] gradients, accumulated */
  const pfpp_tlayer_adamw* ada_adamw_w;         /* [n_layers] or NULL */
  const pfpp_tlayer_adamw* ada_adamw_b;         /* [n_layers] or NULL */
} pfpp_tlayers_args;
int64_t pfpp_tlayers_frag_bytes(int64_t C, int64_t inner);
int64_t pfpp_tlayers_fwd_bytes(int64_t M, int64_t C, int64_t H, int64_t inner);
int64_t pfpp_tlayers_fwd_hout_offset(int64_t M, int64_t C, int64_t H, int64_t inner);
int64_t pfpp_tlayers_bwd_bytes(int64_t M, int64_t C, int64_t H, int64_t inner);
int pfpp_tlayers_fwd(const pfpp_tlayers_args* args, int32_t layer_lo, int32_t layer_hi, pfpp_stream_t stream);
/* layers layer_hi-1 ... layer_lo; side = stream of the weight-gradient GEMMs (NULL: the main stream) */
int pfpp_tlayers_bwd(const pfpp_tlayers_args* args, int32_t layer_lo, int32_t layer_hi, pfpp_stream_t stream, pfpp_stream_t side);

/* The same blocks in EVAL mode for a compacted fragment list (the sampler / auto_aggl step, denoiser.py:172-185 ->
 * DenoiserTransformer.forward, denoiser_transformer.py:187-196): h [M, C] fp32 is updated in place through n_layers blocks; the
 * launches are those of pfpp_layernorm_grouped_split / pfpp_layernorm_split, pfpp_gemm (packed weights: planes of scale * W, the
 * GEGLU pair interleaved 32 value / 32 gate rows, gate applied in the epilogue), pfpp_attn_blockdiag_split and pfpp_attn_dense_split,
 * enqueued from one call.  norm / att [M, C] and u [M, inner] are scratch planes, qkv [M, 3C] scratch fp32 (caller-owned);
 * split_ws / split_cnt: the split-K workspace of pfpp_gemm (see pfpp_gemm_args).                                              */
/* fhi / flo (optional, NULL = absent): the same planes FRAGMENT-BLOCKED for the few-token kernels (pfpp_layernorm_linear_small): block
 * (row tile r = n / 32, k-step s = k / 16) is 1 KB = 64 x 8 halfs, entry 32 (k % 16 / 8) + n % 32 holds W[n][16 s + 8 (k % 16 / 8) .. + 8) — the
 * B operand of one v_mfma_f32_32x32x16_f16 as the 64 lanes hold it; blocks ordered [r][s].  N % 32 == 0, K % 16 == 0, no row padding. */
typedef struct pfpp_pw { const float* f32; const void* hi; const void* lo; float scale; int64_t ldw; const void* fhi; const void* flo; } pfpp_pw;
typedef struct pfpp_elayer_params {
  pfpp_pw qkv1, o1, qkv2, o2, ff1, ff2;            /* [3C,C], [C,C], [3C,C], [C,C], [2 inner (interleaved), C], [C, inner] */
  const float *bo1, *bo2, *g3, *b3, *bff1, *bff2;
} pfpp_elayer_params;
typedef struct pfpp_tlayers_eval_args {
  int32_t n_layers;
  const pfpp_elayer_params* layers;
  int64_t M, C, H, L, inner, Fv, B;
  float* h;
  const float* mods;                               /* [2 n_layers, B, 2C] */
  const int32_t* frag_b; const int32_t *seq_off, *seq_len;
  int64_t n_seq, max_len;
  float att_scale;
  int32_t single_pass;                             /* 1: PFPP_GEMM_F16 on the GEMMs (perf mode) */
  pfpp_planes norm, att, u; float* qkv;
  float* split_ws; int64_t split_ws_bytes; int32_t* split_cnt; int64_t split_cnt_len;
  int64_t lnlin_max_rows;                          /* M <= this: LayerNorm + the following linear as one launch (pfpp_layernorm_linear_small) */
  int32_t wd_gemm;                                 /* 1: above lnlin_max_rows the qkv / out / second feed-forward linears through pfpp_gemm_wd */
} pfpp_tlayers_eval_args;
int pfpp_tlayers_eval(const pfpp_tlayers_eval_args* args, pfpp_stream_t stream);
/* LayerNorm fused into the linear layer that follows it, for small token counts (one puzzle in flight): y = LN(x) . W^T (+ bias), the
 * normalised rows never reach HBM (MyAdaLayerNorm / nn.LayerNorm + to_q|k|v resp. the GEGLU projection, attention.py:21-25,77-90, eval
 * mode).  mod [B, 2C] (scale | shift) with group_batch / group_rows as in pfpp_layernorm_grouped, or gamma / beta.  Plain form: out
 * [M, ldc] fp32.  GEGLU form (u_planes set; w = packed 32 value | 32 gate rows, bias packed likewise): u = (v + b_v) * gelu(g + b_g)
 * as planes [M, ldu], N = 2 * inner.  C = 512, N % 64 == 0.  LayerNorm arithmetic = pfpp_layernorm*; the contraction is summed in a
 * different (fixed) order than pfpp_gemm's: equal to the two-launch path to fp32 rounding.  pfpp_tlayers_eval uses it for M <= lnlin_max_rows
 * (pfpp_hip passes 2048: up to there the few-token kernels beat the tiled GEMMs in the auto_aggl loop; 0 = never).                                                                                                  */
/* Token embedding for few tokens, one launch (denoiser_transformer.py:117-135, 150-156, 173-185; EmbedderNerf, utils/model_utils.py:68-69):
 * tok[(f, l), :] = shape_embedding([latent | PE(xyz) | PE(scale)]) + param_fc(PE(x_f)) + ref_part_emb[ref_f] + pe[frag_pos_f] for the n listed
 * fragments (slot: listed fragment -> slot of latent / xyz / scale / x / ref_part, or NULL).  w_cat = the CONCATENATED weight
 * [W_shape (148 columns) | W_param (147) | 0 ... ] [C, 320] with its fragment-blocked planes, bias = shape bias + param bias.  Replaces
 * pfpp_token_features + two pfpp_gemm + pfpp_token_combine for small n L (pfpp_hip: <= 2,048 tokens); equal to them to fp32 rounding
 * (both linear layers share one accumulation).  L >= 11, C % 32 == 0.                                                                */
int pfpp_embed_tokens_small(const float* latent, const float* xyz, const float* scale, const float* x, const int32_t* slot,
                            const pfpp_pw* w_cat, const float* bias, const float* ref_emb, const uint8_t* ref_part, const float* pe,
                            const int32_t* frag_pos, float* tok, int64_t n, int64_t L, int64_t C, pfpp_stream_t stream);
/* Plane GEMM for few rows: out [M, ldc] = A . W^T / (A.scale * w.scale) + bias + residual, A = planes [M, lda] of a [M, K] operand, w
 * with its fragment-blocked planes (fhi / flo).  The out-projections of both attentions and the second feed-forward linear with their
 * residual adds (attention.py:77-90), eval mode; out may be the residual (in place).  K % 512 == 0, N % 32 == 0.  Same products as
 * pfpp_gemm's split-f16 path, one k-ordered chain per output (equal to it to fp32 rounding); pfpp_tlayers_eval uses it for
 * M <= lnlin_max_rows.                                                                                                              */
int pfpp_gemm_small(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr, float* out,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream);
/* The same operation above the few-token range, weights never staged through LDS (csrc/gemm_wd.hip): out [M, ldc] = (A . W^T) / (A.scale *
 * w.scale) + bias + residual with A = row-major planes [M, lda] and w's fragment-blocked planes read straight into the matrix operands
 * (attn.to_q|k|v / attn.to_out[0] / ff.net[2] of attention.py:77-90, eval mode: static weights).  N % 128 == 0, K % 32 == 0, 16-byte
 * aligned rows; out may be the residual.  Bit-identical to pfpp_gemm's split-f16 plane path on the same operands (same products, same
 * order, same epilogue); pfpp_tlayers_eval uses it for M > lnlin_max_rows when args.wd_gemm is set.                                      */
int pfpp_gemm_wd(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr, float* out,
                 int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream);
int pfpp_gemm_wd_supported(int64_t M, int64_t N, int64_t K);
/* the same in the single-pass fp16 mode (hi planes only, one matrix instruction per product: PFPP_GEMM_F16 of pfpp_gemm — BASELINE configs[4]'s
 * "fp16 MFMA" perf mode, never a parity mode); bit-identical to pfpp_gemm's single-pass plane path */
int pfpp_gemm_wd_f16(const pfpp_planes* A, int64_t lda, const pfpp_pw* w, const float* bias, const float* residual, int64_t ldr, float* out,
                     int64_t ldc, int64_t M, int64_t N, int64_t K, pfpp_stream_t stream);
/* Fragment-blocked copies of row-major weight planes w [N, ldw] (N x K used), one launch for up to PFPP_REBLOCK_MAX weights: fhi / flo
 * receive the layout of pfpp_pw.fhi / flo of W (transposed == 0: N % 32 == 0, K % 16 == 0) or of W^T [K, N] (transposed == 1: the operand
 * of dX = dY . W for pfpp_gemm_wd; N % 64 == 0, K % 64 == 0).  Training weights change every step: pfpp_tlayers_fwd / _bwd make the
 * copies of a layer where they read them (args.frag_ws), so that a copy can never be stale.                                         */
#define PFPP_REBLOCK_MAX 32
typedef struct pfpp_reblock_job { pfpp_planes w; int64_t N, K, ldw; void *fhi, *flo; int32_t transposed; } pfpp_reblock_job;
int pfpp_reblock_planes(const pfpp_reblock_job* jobs, int32_t n, pfpp_stream_t stream);
int pfpp_layernorm_linear_small(const float* x, const float* mod, int64_t ld_mod, const float* gamma, const float* beta,
                                const int32_t* group_batch, int64_t group_rows, const pfpp_pw* w, const float* bias, float* out,
                                int64_t ldc, const pfpp_planes* u_planes, int64_t ldu, int64_t M, int64_t N, int64_t C, float eps,
                                pfpp_stream_t stream);

/* ---- a15: the two output heads as one launch each way -------------------------------------------------------
 * DenoiserTransformer._out (denoiser_transformer.py:138-147) after the mean over the L latent points: pooled [R, C] ->
 * mlp_out_trans / mlp_out_rot (Linear(C,C) - SiLU - Linear(C,C/2) - SiLU - Linear(C/2, 3 | 4), :58-61) -> out[row, 0:3] |
 * out[row, 3:7], row = slot ? slot[r] : r (the scatter back to the padded fragment slots).  a0 / v0 [2, R, C] and a1 / v1
 * [2, R, C/2] (pre-activations and SiLU values of both heads, trans first) are saved for the backward when given (all or none).
 * Weights: split-f16 planes of scale * W (row-major [out, in]) for the two wide layers, fp32 for the last.  C = 512.
 * pfpp_heads_bwd: from dout (unscaled; row r of the heads is dout[slot ? slot[r] : r, 0:7]) it writes da0 [2, R, C], da1 [2, R, C/2] (the dY operands of the four wide
 * weight-gradient GEMMs, which stay with pfpp_gemm_grad_group), accumulates dW4 / db4 / db2 / db0 of both heads with atomics,
 * writes each head's share of d pooled to dp [2, R, C] and, when dx is given, dx [(r, l), :] = (dp[0] + dp[1])[r, :] / L
 * (the backward of the mean pool, :139-142).  grad_scale: power of two lifting the gradient planes into the fp16 range.   */
typedef struct pfpp_head_params {
  pfpp_planes w0, w2;                         /* [C, C], [C/2, C] */
  const float *w4, *b0, *b2, *b4;             /* [3 | 4, C/2], [C], [C/2], [3 | 4] */
  pfpp_planes f0, f2;                         /* optional (hi == NULL: absent): w0 / w2 fragment-blocked as pfpp_pw.fhi / flo — static (eval)
                                                 weights; pfpp_heads_fwd then loads the MFMA operands with fully coalesced instructions */
} pfpp_head_params;
typedef struct pfpp_head_grads { float *w4, *b4, *b2, *b0; } pfpp_head_grads;
int pfpp_heads_fwd(const float* pooled, const pfpp_head_params* trans, const pfpp_head_params* rot, int64_t R, int64_t C,
                   float* a0, float* v0, float* a1, float* v1, float* out, const int32_t* slot, int64_t ldo,
                   pfpp_stream_t stream);
int pfpp_heads_bwd(const float* dout, const int32_t* slot, const pfpp_head_params* trans, const pfpp_head_params* rot, int64_t R, int64_t C,
                   const float* a0, const float* v0, const float* a1, const float* v1, float* da0, float* da1, float* dp,
                   const pfpp_head_grads* g_trans, const pfpp_head_grads* g_rot, float grad_scale, float* dx, int64_t L,
                   pfpp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PFPP_H_ */
