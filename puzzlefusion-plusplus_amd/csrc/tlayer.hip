// Host-side sequencing of the DenoiserTransformer's blocks for the training step (a17): pfpp_tlayers_fwd / pfpp_tlayers_bwd enqueue the
// launches of a range of transformer layers from C++, in the order — and with the arguments — that pfpp_hip/train.py's
// _forward_layers_planes / _backward_layers_planes issue them one ctypes call at a time (that Python sequence stays as the cross-check:
// PFPP_TRAIN_CSEQ=0).  No arithmetic of its own: every launch goes through the library's public entry points.
//
// Reference: EncoderLayer.forward (denoiser/model/modules/attention.py:74-92: AdaLN -> self-attention (block-diagonal mask) -> +res ->
// AdaLN -> global attention -> +res -> LayerNorm -> GEGLU feed-forward -> +res) for the layers of DenoiserTransformer.forward
// (denoiser_transformer.py:187-196) in train mode, and its autograd in Denoiser.training_step (denoiser.py:128-145).
//
// Why: the iteration issues ~360 launches; ~240 of them belong to the six blocks.  From Python (argument marshalling + one foreign call
// each) the host needs 6.3 ms to enqueue an iteration the GPU runs in 6.7 ms — any further GPU-side gain would be
// hidden behind the enqueue.  From here a launch costs what hipLaunchKernel costs.
#include <stdlib.h>

#include <vector>

#include "pfpp_common.h"

namespace {

struct EventRing {
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  hipEvent_t get() {
    if (ev.empty()) {
      ev.resize(64);
      for (auto& e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    hipEvent_t e = ev[next];
    next = (next + 1) % ev.size();
    return e;
  }
};
EventRing g_ring;      // re-recording an event does not disturb waits already enqueued on it; 64 keeps us far from any doubt

// everything queued on `producer` so far happens before whatever `consumer` is given next
int order_after(hipStream_t consumer, hipStream_t producer) {
  hipEvent_t e = g_ring.get();
  if (!e || hipEventRecord(e, producer) != hipSuccess || hipStreamWaitEvent(consumer, e, 0) != hipSuccess) {
    pfpp::set_error("pfpp_tlayers: hipEventRecord / hipStreamWaitEvent failed");
    return PFPP_EHIP;
  }
  return PFPP_OK;
}

// A plane buffer of the backward chain that a weight-gradient GEMM on the side stream reads while the chain moves on: before the chain
// writes it again (one or two layers later) it waits for that read.  Dedicated events (the ring's are re-recorded by other producers).
struct Slot {
  char* buf = nullptr;
  hipEvent_t read_done = nullptr;
  bool pending = false;
};
enum { SLOT_DZ0, SLOT_DZ1, SLOT_QA0, SLOT_QA1, SLOT_QB0, SLOT_QB1, SLOT_DY0, SLOT_DY1, SLOT_DY2, SLOT_DY3, SLOT_DH0, SLOT_DH1, N_SLOTS };
Slot g_slots[N_SLOTS];      // one backward at a time per process (the engine's backward is not re-entrant either)

int slot_acquire(Slot& s, char* buf, hipStream_t main_s) {
  if (s.pending && s.buf == buf && hipStreamWaitEvent(main_s, s.read_done, 0) != hipSuccess) {
    pfpp::set_error("pfpp_tlayers_bwd: hipStreamWaitEvent failed");
    return PFPP_EHIP;
  }
  s.buf = buf;
  s.pending = false;
  return PFPP_OK;
}
int slot_read_on(Slot& s, hipStream_t side_s) {
  if (!s.read_done && hipEventCreateWithFlags(&s.read_done, hipEventDisableTiming) != hipSuccess) {
    pfpp::set_error("pfpp_tlayers_bwd: hipEventCreate failed");
    return PFPP_EHIP;
  }
  if (hipEventRecord(s.read_done, side_s) != hipSuccess) {
    pfpp::set_error("pfpp_tlayers_bwd: hipEventRecord failed");
    return PFPP_EHIP;
  }
  s.pending = true;
  return PFPP_OK;
}

#define TL_CALL(expr) do { const int rc_ = (expr); if (rc_ != PFPP_OK) return rc_; } while (0)

inline float* f32_at(void* base, int64_t off) { return reinterpret_cast<float*>(static_cast<char*>(base) + off); }
inline pfpp_planes planes_at(void* base, int64_t off, int64_t halfs, float scale) {
  pfpp_planes p;
  p.hi = static_cast<char*>(base) + off;
  p.lo = static_cast<char*>(base) + off + halfs * 2;
  p.scale = scale;
  return p;
}

// byte offsets of one layer's saved activations inside its slice of the forward arena
struct FwdLayout {
  int64_t n1, qkv1, att1, y1, n2, qkv2, att2, att2p, lse, y2, n3, z, u, hout, total;
};
FwdLayout fwd_layout(int64_t M, int64_t C, int64_t H, int64_t inner) {
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  FwdLayout f;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t at = o; o += up(bytes); return at; };
  f.n1 = take(M * C * 4);          // planes: hi + lo = 4 bytes per element
  f.qkv1 = take(M * 3 * C * 4);
  f.att1 = take(M * C * 4);
  f.y1 = take(M * C * 4);          // out-projection (+ bias), then h1 in place
  f.n2 = take(M * C * 4);
  f.qkv2 = take(M * 3 * C * 4);
  f.att2 = take(M * C * 4);
  f.att2p = take(M * C * 4);
  f.lse = take(M * H * 4);
  f.y2 = take(M * C * 4);
  f.n3 = take(M * C * 4);
  f.z = take(M * 2 * inner * 4);
  f.u = take(M * inner * 4);
  f.hout = take(M * C * 4);
  f.total = o;
  return f;
}

int gemm_pl(const pfpp_planes& A, const pfpp_planes& W, float* Cout, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
            bool a_km, bool w_km, const float* bias, const float* residual, bool accumulate, float* colsum, float* ws, int64_t ws_bytes,
            int single_pass, pfpp_stream_t st, int splits = 0, pfpp_slab_job* defer = nullptr, int variant = 0) {
  pfpp_gemm_planes_args a = {};
  a.splits = splits;
  a.variant = variant;
  a.defer = defer;
  a.a_hi = A.hi; a.a_lo = A.lo; a.w_hi = W.hi; a.w_lo = W.lo;
  a.C = Cout; a.bias = bias; a.residual = residual;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = residual ? ldc : 0;
  a.a_kmajor = a_km; a.w_kmajor = w_km;
  a.act = PFPP_ACT_NONE;
  a.accumulate = accumulate;
  a.alpha = 1.0f / (A.scale * W.scale);
  a.single_pass = single_pass;
  a.ws = ws; a.ws_bytes = ws_bytes;
  if (colsum) { a.colsum = colsum; a.colsum_alpha = 1.0f / A.scale; }
  return pfpp_gemm_planes(&a, st);
}

// ---- weights straight into the matrix operands (csrc/gemm_wd.hip): the fragment-blocked copies of the layer's qkv1 / o1 / qkv2 / o2 / ff2
// planes (transposed: of their transposes, the operands of the input gradients) are made in frag_ws right before they are read, on the
// stream that reads them — a copy can never be stale, whoever updated the weights
struct LayerFrag { pfpp_pw qkv1, o1, qkv2, o2, ff2; };
int64_t frag_halfs(int64_t C, int64_t inner) { return 2 * 3 * C * C + 2 * C * C + C * inner; }      // per plane
bool wd_shapes_ok(int64_t M, int64_t C, int64_t inner) {
  return C % 128 == 0 && inner % 128 == 0 && pfpp_gemm_wd_supported(M, C, C) && pfpp_gemm_wd_supported(M, 3 * C, C) &&
         pfpp_gemm_wd_supported(M, C, inner) && pfpp_gemm_wd_supported(M, C, 3 * C) && pfpp_gemm_wd_supported(M, inner, C);
}
// jobs of one layer's five weights into the scratch slice `slice` (0 = the only one); returns the number of jobs appended
int layer_jobs(const pfpp_tlayers_args* a, const pfpp_tlayer_params& w, bool transposed, int slice, LayerFrag* out, pfpp_reblock_job* jobs) {
  const int64_t C = a->C, inner = a->inner;
  _Float16* hi = static_cast<_Float16*>(a->frag_ws) + (int64_t)slice * 2 * frag_halfs(C, inner);
  _Float16* lo = hi + frag_halfs(C, inner);
  const pfpp_planes* src[5] = {&w.qkv1, &w.o1, &w.qkv2, &w.o2, &w.ff2};
  const int64_t N[5] = {3 * C, C, 3 * C, C, C}, K[5] = {C, C, C, C, inner};
  pfpp_pw* dst[5] = {&out->qkv1, &out->o1, &out->qkv2, &out->o2, &out->ff2};
  int64_t off = 0;
  for (int i = 0; i < 5; ++i) {
    jobs[i].w = *src[i]; jobs[i].N = N[i]; jobs[i].K = K[i]; jobs[i].ldw = K[i];
    jobs[i].fhi = hi + off; jobs[i].flo = lo + off; jobs[i].transposed = transposed ? 1 : 0;
    *dst[i] = pfpp_pw{nullptr, src[i]->hi, src[i]->lo, src[i]->scale, K[i], hi + off, lo + off};
    off += N[i] * K[i];
  }
  return 5;
}
int reblock_layer(const pfpp_tlayers_args* a, const pfpp_tlayer_params& w, bool transposed, LayerFrag* out, pfpp_stream_t st) {
  pfpp_reblock_job jobs[5];
  layer_jobs(a, w, transposed, 0, out, jobs);
  return pfpp_reblock_planes(jobs, 5, st);
}
// all layers of [lo, hi) in one launch (needs (hi - lo) scratch slices and 5 (hi - lo) <= PFPP_REBLOCK_MAX jobs); fr[i - lo] = layer i's views
constexpr int MAX_BATCHED_LAYERS = PFPP_REBLOCK_MAX / 5;
bool can_batch(const pfpp_tlayers_args* a, int lo, int hi) {
  return hi - lo > 1 && hi - lo <= MAX_BATCHED_LAYERS && a->frag_ws_bytes >= (int64_t)(hi - lo) * 2 * frag_halfs(a->C, a->inner) * 2;
}
int reblock_range(const pfpp_tlayers_args* a, int lo, int hi, bool transposed, LayerFrag* fr, pfpp_stream_t st) {
  pfpp_reblock_job jobs[PFPP_REBLOCK_MAX];
  int n = 0;
  for (int i = lo; i < hi; ++i) n += layer_jobs(a, a->layers[i], transposed, i - lo, &fr[i - lo], jobs + n);
  return pfpp_reblock_planes(jobs, n, st);
}

}  // namespace

extern "C" int64_t pfpp_tlayers_frag_bytes(int64_t C, int64_t inner) { return 2 * frag_halfs(C, inner) * 2; }

extern "C" int64_t pfpp_tlayers_fwd_bytes(int64_t M, int64_t C, int64_t H, int64_t inner) { return fwd_layout(M, C, H, inner).total; }

extern "C" int64_t pfpp_tlayers_fwd_hout_offset(int64_t M, int64_t C, int64_t H, int64_t inner) { return fwd_layout(M, C, H, inner).hout; }

extern "C" int pfpp_tlayers_fwd(const pfpp_tlayers_args* a, int32_t layer_lo, int32_t layer_hi, pfpp_stream_t stream) {
  PFPP_REQUIRE(a && a->layers && a->fwd_arena && a->h_in && a->mods && a->frag_b && a->seq_off && a->seq_len, "null pointer");
  PFPP_REQUIRE(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= a->n_layers, "bad layer range");
  const int64_t M = a->M, C = a->C, H = a->H, L = a->L, inner = a->inner, dh = C / H;
  PFPP_REQUIRE(M > 0 && C > 0 && H > 0 && C % H == 0 && inner > 0 && L > 0 && M == a->Fv * L, "bad sizes");
  const FwdLayout lo = fwd_layout(M, C, H, inner);
  PFPP_REQUIRE(a->fwd_layer_bytes >= lo.total, "fwd_layer_bytes smaller than pfpp_tlayers_fwd_bytes()");
  const float eps = 1e-5f;
  const int64_t ld_mod = 2 * C;
  const bool wd = a->frag_ws != nullptr && wd_shapes_ok(M, C, inner);
  if (wd) PFPP_REQUIRE(a->frag_ws_bytes >= pfpp_tlayers_frag_bytes(C, inner), "frag_ws_bytes smaller than pfpp_tlayers_frag_bytes()");
  // a linear of the block: out = A . W^T + bias (+ residual) — weights straight into the matrix operands, or the tiled kernel (bit-identical)
  LayerFrag frs[MAX_BATCHED_LAYERS], fr;
  const bool batched = wd && can_batch(a, layer_lo, layer_hi);
  if (batched) TL_CALL(reblock_range(a, layer_lo, layer_hi, false, frs, stream));
  auto lin = [&](const pfpp_planes& A, const pfpp_planes& W, const pfpp_pw& F, float* out, int64_t N, int64_t K, const float* bias,
                 const float* residual) -> int {
    if (wd) { const pfpp_planes Ac = A; return pfpp_gemm_wd(&Ac, K, &F, bias, residual, N, out, N, M, N, K, stream); }
    return gemm_pl(A, W, out, M, N, K, K, K, N, false, false, bias, residual, false, nullptr, a->ws_main, a->ws_bytes, 0, stream);
  };
  for (int i = layer_lo; i < layer_hi; ++i) {
    const pfpp_tlayer_params& w = a->layers[i];
    if (batched) fr = frs[i - layer_lo];
    else if (wd) TL_CALL(reblock_layer(a, w, false, &fr, stream));
    char* base = static_cast<char*>(a->fwd_arena) + (int64_t)i * a->fwd_layer_bytes;
    // the residual stream entering the layer: the tokens for layer 0, the previous layer's output otherwise
    float* h = i == 0 ? a->h_in : f32_at(static_cast<char*>(a->fwd_arena) + (int64_t)(i - 1) * a->fwd_layer_bytes, lo.hout);
    const float* mod1 = a->mods + (int64_t)(2 * i) * a->B * ld_mod;
    const float* mod2 = a->mods + (int64_t)(2 * i + 1) * a->B * ld_mod;
    pfpp_planes n1 = planes_at(base, lo.n1, M * C, 1.0f), att1 = planes_at(base, lo.att1, M * C, 1.0f);
    pfpp_planes n2 = planes_at(base, lo.n2, M * C, 1.0f), att2p = planes_at(base, lo.att2p, M * C, 1.0f);
    pfpp_planes n3 = planes_at(base, lo.n3, M * C, 1.0f), u = planes_at(base, lo.u, M * inner, 1.0f);
    float *qkv1 = f32_at(base, lo.qkv1), *y1 = f32_at(base, lo.y1), *qkv2 = f32_at(base, lo.qkv2), *att2 = f32_at(base, lo.att2);
    float *lse = f32_at(base, lo.lse), *y2 = f32_at(base, lo.y2), *z = f32_at(base, lo.z), *hout = f32_at(base, lo.hout);
    // ---- self attention (attention.py:77-80)
    if (i == 0 && a->p_tok > 0.0f)        // PositionalEncoding's token dropout rides in the first LayerNorm (in place on the tokens)
      TL_CALL(pfpp_dropout_layernorm_p(h, nullptr, h, nullptr, mod1, ld_mod, nullptr, nullptr, a->frag_b, L, 1, M, C, eps, a->p_tok, a->seed, 0,
                                       &n1, stream));
    else
      TL_CALL(pfpp_layernorm_grouped_split(h, n1.hi, n1.lo, mod1, ld_mod, a->frag_b, L, M, C, eps, stream));
    TL_CALL(lin(n1, w.qkv1, fr.qkv1, qkv1, 3 * C, C, nullptr, nullptr));
    TL_CALL(pfpp_attn_blockdiag_split(qkv1, att1.hi, att1.lo, a->Fv, L, H, dh, a->att_scale, stream));
    if (a->p_lay > 0.0f) {
      TL_CALL(lin(att1, w.o1, fr.o1, y1, C, C, w.bo1, nullptr));
      TL_CALL(pfpp_dropout_layernorm_p(y1, h, y1, nullptr, mod2, ld_mod, nullptr, nullptr, a->frag_b, L, 1, M, C, eps, a->p_lay, a->seed,
                                       (uint32_t)(1 + 3 * i), &n2, stream));
    } else {
      TL_CALL(lin(att1, w.o1, fr.o1, y1, C, C, w.bo1, h));
      TL_CALL(pfpp_layernorm_grouped_split(y1, n2.hi, n2.lo, mod2, ld_mod, a->frag_b, L, M, C, eps, stream));
    }
    // ---- global attention (attention.py:82-85)
    TL_CALL(lin(n2, w.qkv2, fr.qkv2, qkv2, 3 * C, C, nullptr, nullptr));
    TL_CALL(pfpp_attn_dense_train_p(qkv2, att2, lse, a->seq_off, a->seq_len, nullptr, 0, a->n_seq, a->max_len, H, dh, a->att_scale, &att2p, stream));
    if (a->p_lay > 0.0f) {
      TL_CALL(lin(att2p, w.o2, fr.o2, y2, C, C, w.bo2, nullptr));
      TL_CALL(pfpp_dropout_layernorm_p(y2, y1, y2, nullptr, nullptr, 0, w.g3, w.b3, nullptr, 1, 1, M, C, eps, a->p_lay, a->seed,
                                       (uint32_t)(2 + 3 * i), &n3, stream));
    } else {
      TL_CALL(lin(att2p, w.o2, fr.o2, y2, C, C, w.bo2, y1));
      TL_CALL(pfpp_layernorm_split(y2, n3.hi, n3.lo, nullptr, 0, w.g3, w.b3, M, C, 1, eps, stream));
    }
    // ---- GEGLU feed-forward (attention.py:87-90)
    TL_CALL(gemm_pl(n3, w.ff1, z, M, 2 * inner, C, C, C, 2 * inner, false, false, w.bff1, nullptr, false, nullptr, a->ws_main, a->ws_bytes, 0, stream));
    TL_CALL(pfpp_geglu_p(z, nullptr, M, inner, a->p_lay, a->seed, (uint32_t)(3 + 3 * i), &u, stream));
    TL_CALL(lin(u, w.ff2, fr.ff2, hout, C, inner, w.bff2, y2));
  }
  return PFPP_OK;
}

// Backward of layers [layer_lo, layer_hi), last first.  dh [M, C] fp32 is the running gradient of the residual stream (updated in place),
// dhp its planes (grad_scale * dh) — on entry the gradient of layer_hi - 1's output, on exit that of layer_lo's input (for layer_lo == 0:
// dtok [M, C] receives the gradient with respect to the tokens instead, token dropout applied).  Weight / bias gradients go to `side`
// (ordered after what they read; NULL = the main stream), optionally followed by the layer's AdamW update there (a->adamw).
extern "C" int pfpp_tlayers_bwd(const pfpp_tlayers_args* a, int32_t layer_lo, int32_t layer_hi, pfpp_stream_t stream, pfpp_stream_t side) {
  PFPP_REQUIRE(a && a->layers && a->fwd_arena && a->bwd_arena && a->h_in && a->mods && a->frag_b && a->seq_off && a->seq_len && a->dh &&
               a->dhp.hi && a->dhp.lo && a->dmods, "null pointer");
  PFPP_REQUIRE(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= a->n_layers, "bad layer range");
  PFPP_REQUIRE(layer_lo > 0 || a->dtok, "dtok is needed when the range reaches layer 0");
  const int64_t M = a->M, C = a->C, H = a->H, L = a->L, inner = a->inner, dh = C / H;
  const FwdLayout lo = fwd_layout(M, C, H, inner);
  const float eps = 1e-5f, G = a->grad_scale;
  const int64_t ld_mod = 2 * C;
  hipStream_t main_s = pfpp::as_stream(stream);
  hipStream_t side_s = side ? pfpp::as_stream(side) : main_s;
  pfpp_stream_t side_t = side ? side : stream;
  float* ws_side = side ? a->ws_side : a->ws_main;
  // temporaries of the backward chain.  fp32 intermediates (du, dn, datt, dvec) only ever live on the main stream and are reused at
  // once; the plane buffers are also read by weight-gradient GEMMs on the side stream: two (by layer parity) of each, and the chain
  // waits for the side stream's read before it writes one again (Slot)
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  char* tb = static_cast<char*>(a->bwd_arena);
  int64_t o = 0;
  auto take = [&](int64_t bytes) { char* p = tb + o; o += up(bytes); return p; };
  char* du_b = take(M * inner * 4);
  char* dn_b = take(M * C * 4);
  char* datt_b = take(M * C * 4);
  char* dvec_b = take(M * H * 4);
  char* slot_buf[N_SLOTS];
  for (int k = SLOT_DZ0; k <= SLOT_DZ1; ++k) slot_buf[k] = take(M * 2 * inner * 4);
  for (int k = SLOT_QA0; k <= SLOT_QB1; ++k) slot_buf[k] = take(M * 3 * C * 4);
  for (int k = SLOT_DY0; k <= SLOT_DH1; ++k) slot_buf[k] = take(M * C * 4);
  PFPP_REQUIRE(a->bwd_bytes >= o, "bwd_bytes smaller than pfpp_tlayers_bwd_bytes()");

  int cur_layer = 0;
  // PFPP_TRAIN_GROUP_REDUCE=1: the K splits of a block's six weight-gradient GEMMs keep their slabs in their own stretches of the side
  // workspace and are reduced by ONE launch behind the block's last weight gradient (42 -> 12 reduction launches per iteration).
  // Measured (profiles/r04o_ab_group_reduce.txt): 0.04-0.1 ms per iteration SLOWER — 92 MB of slabs per block are re-read after the
  // other five GEMMs have pushed them out of L2 instead of right behind their GEMM — so every GEMM reduces at once by default.
  static const bool group_reduce = getenv("PFPP_TRAIN_GROUP_REDUCE") && atoi(getenv("PFPP_TRAIN_GROUP_REDUCE")) == 1;
  pfpp_slab_job jobs[PFPP_SLAB_GROUP_MAX];
  int n_jobs = 0;
  int64_t ws_used = 0;
  auto flush_jobs = [&]() -> int {
    const int n = n_jobs;
    n_jobs = 0; ws_used = 0;
    return n ? pfpp_slab_reduce_group(jobs, n, side_t) : PFPP_OK;
  };
  // planes of scale G in slot k, safe to write on the main stream
  auto fresh = [&](int k, int64_t elems, pfpp_planes* out) -> int {
    if (side) TL_CALL(slot_acquire(g_slots[k], slot_buf[k], main_s));
    *out = planes_at(slot_buf[k], 0, elems, G);
    return PFPP_OK;
  };
  // PFPP_TRAIN_DW_GROUP (default 1, round 5): a block's six weight gradients as ONE launch (pfpp_gemm_dw_group, csrc/gemm_pl.hip: every
  // output tile over the whole contraction, no K split, no slabs, no reduction launches) behind the block's last gradient kernel,
  // 2: two launches (feed-forward pair behind the GEGLU backward, the four attention linears at the block's end), 3: two launches for
  // layer 0 only (the last block of the backward: what is left on the side stream when the chain ends is the iteration's tail); 0 = one
  // pfpp_gemm_planes launch (+ slab reduction) per weight as through round 4 (the cross-check of the tests)
  const int dw_group = getenv("PFPP_TRAIN_DW_GROUP") ? atoi(getenv("PFPP_TRAIN_DW_GROUP")) : 1;          // (read per call: the tests switch it)
  const int dw_group_variant = getenv("PFPP_TRAIN_DW_GROUP_VARIANT") ? atoi(getenv("PFPP_TRAIN_DW_GROUP_VARIANT")) : 0;
  pfpp_dw_job dw_jobs[PFPP_DW_GROUP_MAX];
  int dw_slots[PFPP_DW_GROUP_MAX];
  int n_dw = 0;
  auto flush_dw = [&]() -> int {
    if (n_dw == 0) return PFPP_OK;
    if (side) TL_CALL(order_after(side_s, main_s));
    TL_CALL(pfpp_gemm_dw_group(dw_jobs, n_dw, M, dw_group_variant, side_t));
    for (int q = 0; q < n_dw; ++q)
      if (side && dw_slots[q] >= 0) TL_CALL(slot_read_on(g_slots[dw_slots[q]], side_s));
    n_dw = 0;
    return PFPP_OK;
  };
  // dW += dY^T . X, db += colsum(dY): both operands read in place as k-major planes, on the side stream; `k` = the slot dY lives in
  // lab switches of the weight-gradient path, parsed once per call of this function (not per weight): PFPP_LAB_SKIP_DW_LAYERS (timing only,
  // WRONG gradients: no dW for the last k layers to run) applies to the grouped launch too; PFPP_DW_SPLITS only means something with
  // PFPP_TRAIN_DW_GROUP=0 (the grouped launch has no K split)
  const int dw_splits = getenv("PFPP_DW_SPLITS") ? atoi(getenv("PFPP_DW_SPLITS")) : 0;
  const int lab_skip = getenv("PFPP_LAB_SKIP_DW_LAYERS") ? atoi(getenv("PFPP_LAB_SKIP_DW_LAYERS")) : 0;
  // INVARIANT the grouped path relies on: between two flush_dw() calls every dY slot is acquired (fresh) at most ONCE - the slots'
  // read events are recorded at flush time only, so a slot re-acquired and rewritten before the flush would race with the queued
  // read of its first tenant.  Holds today: a block uses each of its slots for one dY, and every block ends in flush_dw().
  auto dw = [&](const pfpp_planes& dyp, int k, const pfpp_planes& xp, int64_t n_out, int64_t n_in, float* gw, float* gb) -> int {
    if (lab_skip > 0 && cur_layer >= a->n_layers - lab_skip) return PFPP_OK;
    if (dw_group && n_out % 8 == 0 && n_in % 8 == 0) {
      for (int q = 0; q < n_dw; ++q)
        if (k >= 0 && dw_slots[q] == k) { pfpp::set_error("pfpp_tlayers_bwd: slot %d queued twice before a flush", k); return PFPP_EINVAL; }
      if (n_dw == PFPP_DW_GROUP_MAX) TL_CALL(flush_dw());
      pfpp_dw_job& q = dw_jobs[n_dw];
      q.dy = dyp; q.x = xp; q.gw = gw; q.gb = gb; q.M = n_out; q.N = n_in;
      dw_slots[n_dw++] = k;
      return PFPP_OK;
    }
    if (side) TL_CALL(order_after(side_s, main_s));
    if (group_reduce) {
      if (n_jobs == PFPP_SLAB_GROUP_MAX) TL_CALL(flush_jobs());
      pfpp_slab_job& q = jobs[n_jobs];
      TL_CALL(gemm_pl(dyp, xp, gw, n_out, n_in, M, n_out, n_in, n_in, true, true, nullptr, nullptr, true, gb, ws_side + ws_used / 4,
                      a->ws_bytes - ws_used, 0, side_t, dw_splits, &q));
      if (q.splits) { ws_used += up((int64_t)q.splits * q.M * (q.N + 1) * 4); ++n_jobs; }
    } else {
      TL_CALL(gemm_pl(dyp, xp, gw, n_out, n_in, M, n_out, n_in, n_in, true, true, nullptr, nullptr, true, gb, ws_side, a->ws_bytes, 0, side_t, dw_splits));
    }
    if (side && k >= 0) TL_CALL(slot_read_on(g_slots[k], side_s));
    return PFPP_OK;
  };
  const bool wd = a->frag_ws != nullptr && wd_shapes_ok(M, C, inner);
  if (wd) PFPP_REQUIRE(a->frag_ws_bytes >= pfpp_tlayers_frag_bytes(C, inner), "frag_ws_bytes smaller than pfpp_tlayers_frag_bytes()");
  LayerFrag frs[MAX_BATCHED_LAYERS], fr;      // blocked planes of the TRANSPOSED weights (fr: of the current layer)
  const bool batched = wd && can_batch(a, layer_lo, layer_hi);
  auto dx = [&](const pfpp_planes& dyp, const pfpp_planes& W, float* out, int64_t n_in, int64_t n_out, const pfpp_pw* Ft = nullptr) -> int {
    if (wd && Ft) return pfpp_gemm_wd(&dyp, n_out, Ft, nullptr, nullptr, 0, out, n_in, M, n_in, n_out, stream);
    // lab: tile variant / K split of the one input gradient that stays on the tiled kernel (the GEGLU projection's, K = 2 inner)
    static const int dxv = getenv("PFPP_TRAIN_DXFF1_VARIANT") ? atoi(getenv("PFPP_TRAIN_DXFF1_VARIANT")) : 0;
    static const int dxs = getenv("PFPP_TRAIN_DXFF1_SPLITS") ? atoi(getenv("PFPP_TRAIN_DXFF1_SPLITS")) : 0;
    return gemm_pl(dyp, W, out, M, n_in, n_out, n_out, n_in, n_in, false, true, nullptr, nullptr, false, nullptr, a->ws_main, a->ws_bytes, 0, stream,
                   dxs, nullptr, dxv);
  };

  pfpp_planes dhp = a->dhp;
  int dhp_slot = -1;                    // the caller's buffer on entry (kept alive by the caller), one of ours afterwards
  for (int k = SLOT_DH0; k <= SLOT_DH1; ++k)
    if (a->dhp.hi == slot_buf[k]) dhp_slot = k;      // a continued range: the previous call's dhp_out
  if (batched) TL_CALL(reblock_range(a, layer_lo, layer_hi, true, frs, stream));      // (before any AdamW of this call: dX needs the forward's weights)
  for (int i = layer_hi - 1; i >= layer_lo; --i) {
    cur_layer = i;
    const pfpp_tlayer_params& w = a->layers[i];
    const pfpp_tlayer_grads& g = a->grads[i];
    if (batched) fr = frs[i - layer_lo];
    else if (wd) TL_CALL(reblock_layer(a, w, true, &fr, stream));
    char* base = static_cast<char*>(a->fwd_arena) + (int64_t)i * a->fwd_layer_bytes;
    const float* h0 = i == 0 ? a->h_in : f32_at(static_cast<char*>(a->fwd_arena) + (int64_t)(i - 1) * a->fwd_layer_bytes, lo.hout);
    const float* mod1 = a->mods + (int64_t)(2 * i) * a->B * ld_mod;
    const float* mod2 = a->mods + (int64_t)(2 * i + 1) * a->B * ld_mod;
    float* dmod1 = a->dmods + (int64_t)(2 * i) * a->B * ld_mod;
    float* dmod2 = a->dmods + (int64_t)(2 * i + 1) * a->B * ld_mod;
    const pfpp_planes n1 = planes_at(base, lo.n1, M * C, 1.0f), att1 = planes_at(base, lo.att1, M * C, 1.0f);
    const pfpp_planes n2 = planes_at(base, lo.n2, M * C, 1.0f), att2p = planes_at(base, lo.att2p, M * C, 1.0f);
    const pfpp_planes n3 = planes_at(base, lo.n3, M * C, 1.0f), u = planes_at(base, lo.u, M * inner, 1.0f);
    const float *qkv1 = f32_at(base, lo.qkv1), *h1 = f32_at(base, lo.y1), *qkv2 = f32_at(base, lo.qkv2), *att2 = f32_at(base, lo.att2);
    const float *lse = f32_at(base, lo.lse), *h2 = f32_at(base, lo.y2), *z = f32_at(base, lo.z);
    const int par = i & 1;
    float* du = reinterpret_cast<float*>(du_b);
    float* dn = reinterpret_cast<float*>(dn_b);
    float* datt = reinterpret_cast<float*>(datt_b);
    const int drop = a->p_lay > 0.0f ? 1 : 0;
    // ---- feed-forward (attention.py:87-90)
    TL_CALL(dw(dhp, dhp_slot, u, C, inner, g.ff2_w, g.ff2_b));
    TL_CALL(dx(dhp, w.ff2, du, inner, C, &fr.ff2));
    pfpp_planes dzp, dyp, dqkvp;
    TL_CALL(fresh(SLOT_DZ0 + par, M * 2 * inner, &dzp));
    TL_CALL(pfpp_geglu_bwd_p(z, du, nullptr, M, inner, a->p_lay, a->seed, (uint32_t)(3 + 3 * i), &dzp, stream));
    TL_CALL(dw(dzp, SLOT_DZ0 + par, n3, 2 * inner, C, g.ff1_w, g.ff1_b));
    if (dw_group == 2 || (dw_group == 3 && i == 0)) TL_CALL(flush_dw());      // 3: only the LAST block to run (layer 0) goes out in two launches
    TL_CALL(dx(dzp, w.ff1, dn, C, 2 * inner));
    TL_CALL(fresh(SLOT_DY0 + 2 * par, M * C, &dyp));
    TL_CALL(pfpp_layernorm_bwd_p(h2, dn, nullptr, 0, w.g3, nullptr, 32, 1, a->dh, g.g3, g.b3, 0, M, C, eps, nullptr, a->p_lay, a->seed,
                                 (uint32_t)(2 + 3 * i), drop, &dyp, nullptr, stream));
    // ---- global attention (attention.py:82-85)
    TL_CALL(dw(dyp, SLOT_DY0 + 2 * par, att2p, C, C, g.o2_w, g.o2_b));
    TL_CALL(dx(dyp, w.o2, datt, C, C, &fr.o2));
    TL_CALL(fresh(SLOT_QA0 + par, M * 3 * C, &dqkvp));
    TL_CALL(pfpp_attn_dense_bwd_p(qkv2, att2, datt, lse, reinterpret_cast<float*>(dvec_b), nullptr, a->seq_off, a->seq_len, nullptr, 0, a->n_seq,
                                  a->max_len, H, dh, a->att_scale, &dqkvp, stream));
    TL_CALL(dw(dqkvp, SLOT_QA0 + par, n2, 3 * C, C, g.qkv2_w, nullptr));
    TL_CALL(dx(dqkvp, w.qkv2, dn, C, 3 * C, &fr.qkv2));
    TL_CALL(fresh(SLOT_DY0 + 2 * par + 1, M * C, &dyp));
    TL_CALL(pfpp_layernorm_bwd_p(h1, dn, mod2, ld_mod, nullptr, a->frag_b, L, 1, a->dh, dmod2, dmod2 + C, ld_mod, M, C, eps, nullptr, a->p_lay,
                                 a->seed, (uint32_t)(1 + 3 * i), drop, &dyp, nullptr, stream));
    // ---- self attention (attention.py:77-80)
    TL_CALL(dw(dyp, SLOT_DY0 + 2 * par + 1, att1, C, C, g.o1_w, g.o1_b));
    TL_CALL(dx(dyp, w.o1, datt, C, C, &fr.o1));
    TL_CALL(fresh(SLOT_QB0 + par, M * 3 * C, &dqkvp));
    TL_CALL(pfpp_attn_blockdiag_bwd_p(qkv1, datt, nullptr, a->Fv, L, H, dh, a->att_scale, &dqkvp, stream));
    TL_CALL(dw(dqkvp, SLOT_QB0 + par, n1, 3 * C, C, g.qkv1_w, nullptr));
    TL_CALL(dx(dqkvp, w.qkv1, dn, C, 3 * C, &fr.qkv1));
    if (i > 0) {
      // the updated running gradient is the dY of layer i - 1's second feed-forward linear
      pfpp_planes nxt;
      TL_CALL(fresh(SLOT_DH0 + par, M * C, &nxt));
      TL_CALL(pfpp_layernorm_bwd_p(h0, dn, mod1, ld_mod, nullptr, a->frag_b, L, 1, a->dh, dmod1, dmod1 + C, ld_mod, M, C, eps, nullptr, 0.0f, 0, 0, 0,
                                   nullptr, &nxt, stream));
      dhp = nxt;
      dhp_slot = SLOT_DH0 + par;
    } else if (a->p_tok > 0.0f) {
      TL_CALL(pfpp_layernorm_bwd_dropout(h0, dn, mod1, ld_mod, nullptr, a->frag_b, L, 1, a->dh, dmod1, dmod1 + C, ld_mod, M, C, eps, a->dtok,
                                         a->p_tok, a->seed, 0, stream));
    } else {
      TL_CALL(pfpp_layernorm_bwd(h0, dn, mod1, ld_mod, nullptr, a->frag_b, L, 1, a->dh, dmod1, dmod1 + C, ld_mod, M, C, eps, stream));
      if (a->dtok != a->dh && hipMemcpyAsync(a->dtok, a->dh, (size_t)M * C * 4, hipMemcpyDeviceToDevice, main_s) != hipSuccess)
        return pfpp::check_launch(__func__);
    }
    TL_CALL(flush_dw());
    TL_CALL(flush_jobs());
    if ((a->adamw || a->ada_se) && side) TL_CALL(order_after(side_s, main_s));      // the layer's main-stream kernels are all queued by now
    if (a->adamw && side) {
      // optimizer in the backward: the layer's slice of the flat buffer is final once its weight gradients (side stream) and its
      // LayerNorm gradients (main stream, all queued by now) have run
      const pfpp_tlayer_adamw& ad = a->adamw[i];
      TL_CALL(pfpp_adamw_guarded(ad.p, ad.g, ad.m, ad.v, ad.hi, ad.lo, ad.n, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, a->bc1, a->bc2,
                                 a->opt_g_scale, a->opt_zero_grad, a->overflow, side_t));
    }
    if (a->ada_se && side) {
      // the block's two AdaLN linears (attention.py:21-25): dmods rows 2 i, 2 i + 1 are final (both LayerNorm backward launches of
      // the block have been queued) — bias / weight gradients, the gradient w.r.t. the embedded timestep, then (armed) their AdamW;
      // same launches with the same arguments as pfpp_hip.train's tail issued them for all blocks at once
      const int64_t B = a->B, j = 2 * i;
      const float* dm = a->dmods + j * B * ld_mod;
      TL_CALL(pfpp_colsum(dm, a->ada_gb + j * ld_mod, B, ld_mod, ld_mod, 2, B * ld_mod, ld_mod, 1, side_t));
      pfpp_gemm_grad_args q = {};
      q.A = dm; q.W = a->ada_se + j * B * C; q.C = a->ada_gw + j * ld_mod * C;
      q.M = ld_mod; q.N = C; q.K = B; q.lda = ld_mod; q.ldw = C; q.ldc = C;
      q.a_kmajor = 1; q.w_kmajor = 1; q.accumulate = 1; q.split_k = 0; q.batch = 2;
      q.sA = B * ld_mod; q.sW = B * C; q.sC = ld_mod * C;
      q.a_scale = G; q.w_scale = 1.0f; q.alpha = 1.0f;
      TL_CALL(pfpp_gemm_grad(&q, side_t));
      pfpp_gemm_grad_args d = {};
      d.A = dm; d.W = a->ada_w + j * ld_mod * C; d.C = a->ada_dse + j * B * C;
      d.M = B; d.N = C; d.K = ld_mod; d.lda = ld_mod; d.ldw = C; d.ldc = C;
      d.a_kmajor = 0; d.w_kmajor = 1; d.accumulate = 0; d.split_k = 0; d.batch = 2;
      d.sA = B * ld_mod; d.sW = ld_mod * C; d.sC = B * C;
      d.a_scale = G; d.w_scale = 1.0f; d.alpha = 1.0f;
      TL_CALL(pfpp_gemm_grad(&d, side_t));
      if (a->adamw && a->ada_adamw_w && a->ada_adamw_b) {
        const pfpp_tlayer_adamw* ads[2] = {&a->ada_adamw_w[i], &a->ada_adamw_b[i]};
        for (const pfpp_tlayer_adamw* ad : ads)
          TL_CALL(pfpp_adamw_guarded(ad->p, ad->g, ad->m, ad->v, ad->hi, ad->lo, ad->n, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay,
                                     a->bc1, a->bc2, a->opt_g_scale, a->opt_zero_grad, a->overflow, side_t));
      }
    }
  }
  if (a->dhp_out) *a->dhp_out = dhp;
  return PFPP_OK;
}

extern "C" int64_t pfpp_tlayers_bwd_bytes(int64_t M, int64_t C, int64_t H, int64_t inner) {
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  return up(M * inner * 4) + 2 * up(M * C * 4) + up(M * H * 4) + 2 * up(M * 2 * inner * 4) + 4 * up(M * 3 * C * 4) + 6 * up(M * C * 4);
}

// ------------------------------------------------------------------------------------------------------------------------
// eval mode (the sampler / auto_aggl step on a compacted fragment list): the same blocks as pfpp_hip.denoiser.
// denoiser_forward_compact issues them — LayerNorm -> qkv -> per-fragment attention -> out-projection (+ residual, in place) ->
// LayerNorm -> qkv -> ragged dense attention -> out-projection -> LayerNorm -> GEGLU GEMM (packed weights, gate in the epilogue)
// -> second feed-forward linear (+ residual) — enqueued from one call.  One puzzle in flight is ~100 launches of 5-17 us per DDPM
// step: issued from Python the step was host-bound (1.41 ms enqueue against 1.18 ms of GPU time, tools/diag/graph_time.py).
// ------------------------------------------------------------------------------------------------------------------------
namespace {
int gemm_ev(const pfpp_planes& A, const pfpp_pw& W, float* Cout, const pfpp_planes* Cp, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc,
            const float* bias, const float* residual, int act, int precision, const pfpp_tlayers_eval_args* a, pfpp_stream_t st) {
  pfpp_gemm_args g = {};
  g.a_hi = A.hi; g.a_lo = A.lo;
  g.W = W.f32; g.w_hi = W.hi; g.w_lo = W.lo;
  g.C = Cp ? nullptr : Cout;
  g.c_hi = Cp ? Cp->hi : nullptr; g.c_lo = Cp ? Cp->lo : nullptr;
  g.bias = bias; g.residual = residual;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldw = W.ldw; g.ldc = ldc; g.ldr = residual ? ldc : 0;
  g.act = act; g.batch = 1; g.zdiv = 1; g.precision = precision;
  g.alpha = 1.0f / W.scale;
  g.split_ws = a->split_ws; g.split_ws_bytes = a->split_ws_bytes; g.split_cnt = a->split_cnt; g.split_cnt_len = a->split_cnt_len;
  return pfpp_gemm(&g, st);
}
}  // namespace

extern "C" int pfpp_tlayers_eval(const pfpp_tlayers_eval_args* a, pfpp_stream_t stream) {
  PFPP_REQUIRE(a && a->layers && a->h && a->mods && a->frag_b && a->seq_off && a->seq_len && a->qkv && a->norm.hi && a->att.hi && a->u.hi,
               "null pointer");
  const int64_t M = a->M, C = a->C, H = a->H, L = a->L, inner = a->inner, dh = C / H;
  PFPP_REQUIRE(M > 0 && C > 0 && H > 0 && C % H == 0 && inner > 0 && L > 0 && M == a->Fv * L && a->n_layers >= 0, "bad sizes");
  const float eps = 1e-5f;
  const int64_t ld_mod = 2 * C;
  const int prec = a->single_pass ? PFPP_GEMM_F16 : PFPP_GEMM_F16X3;
  // few tokens (one puzzle in flight): the LayerNorm rides in the GEMM that consumes it (csrc/lnlin_small.hip) — 18 launches less per step
  const bool fuse_ln = !a->single_pass && C == 512 && M <= a->lnlin_max_rows && (2 * inner) % 64 == 0;
  const bool small = fuse_ln && a->layers[0].o1.fhi != nullptr;      // few tokens: the three residual GEMMs through pfpp_gemm_small as well
  // above the few-token range: the qkv / out / second feed-forward linears with the weights read straight into the matrix operands
  // (csrc/gemm_wd.hip; bit-identical to the tiled kernel)
  const bool wd = !fuse_ln && a->wd_gemm && a->layers[0].o1.fhi != nullptr && pfpp_gemm_wd_supported(M, C, C) &&
                  pfpp_gemm_wd_supported(M, 3 * C, C) && pfpp_gemm_wd_supported(M, C, inner);
  const auto gemm_wd = a->single_pass ? pfpp_gemm_wd_f16 : pfpp_gemm_wd;
  for (int i = 0; i < a->n_layers; ++i) {
    const pfpp_elayer_params& w = a->layers[i];
    const float* mod1 = a->mods + (int64_t)(2 * i) * a->B * ld_mod;
    const float* mod2 = a->mods + (int64_t)(2 * i + 1) * a->B * ld_mod;
    if (fuse_ln) {
      TL_CALL(pfpp_layernorm_linear_small(a->h, mod1, ld_mod, nullptr, nullptr, a->frag_b, L, &w.qkv1, nullptr, a->qkv, 3 * C, nullptr, 0, M,
                                          3 * C, C, eps, stream));
    } else {
      TL_CALL(pfpp_layernorm_grouped_split(a->h, a->norm.hi, a->norm.lo, mod1, ld_mod, a->frag_b, L, M, C, eps, stream));
      if (wd) TL_CALL(gemm_wd(&a->norm, C, &w.qkv1, nullptr, nullptr, 0, a->qkv, 3 * C, M, 3 * C, C, stream));
      else TL_CALL(gemm_ev(a->norm, w.qkv1, a->qkv, nullptr, M, 3 * C, C, C, 3 * C, nullptr, nullptr, PFPP_ACT_NONE, prec, a, stream));
    }
    TL_CALL(pfpp_attn_blockdiag_split(a->qkv, a->att.hi, a->att.lo, a->Fv, L, H, dh, a->att_scale, stream));
    if (small) TL_CALL(pfpp_gemm_small(&a->att, C, &w.o1, w.bo1, a->h, C, a->h, C, M, C, C, stream));
    else if (wd) TL_CALL(gemm_wd(&a->att, C, &w.o1, w.bo1, a->h, C, a->h, C, M, C, C, stream));
    else TL_CALL(gemm_ev(a->att, w.o1, a->h, nullptr, M, C, C, C, C, w.bo1, a->h, PFPP_ACT_NONE, prec, a, stream));
    if (fuse_ln) {
      TL_CALL(pfpp_layernorm_linear_small(a->h, mod2, ld_mod, nullptr, nullptr, a->frag_b, L, &w.qkv2, nullptr, a->qkv, 3 * C, nullptr, 0, M,
                                          3 * C, C, eps, stream));
    } else {
      TL_CALL(pfpp_layernorm_grouped_split(a->h, a->norm.hi, a->norm.lo, mod2, ld_mod, a->frag_b, L, M, C, eps, stream));
      if (wd) TL_CALL(gemm_wd(&a->norm, C, &w.qkv2, nullptr, nullptr, 0, a->qkv, 3 * C, M, 3 * C, C, stream));
      else TL_CALL(gemm_ev(a->norm, w.qkv2, a->qkv, nullptr, M, 3 * C, C, C, 3 * C, nullptr, nullptr, PFPP_ACT_NONE, prec, a, stream));
    }
    TL_CALL(pfpp_attn_dense_split(a->qkv, a->att.hi, a->att.lo, a->seq_off, a->seq_len, nullptr, 0, a->n_seq, a->max_len, H, dh, a->att_scale,
                                  stream));
    if (small) TL_CALL(pfpp_gemm_small(&a->att, C, &w.o2, w.bo2, a->h, C, a->h, C, M, C, C, stream));
    else if (wd) TL_CALL(gemm_wd(&a->att, C, &w.o2, w.bo2, a->h, C, a->h, C, M, C, C, stream));
    else TL_CALL(gemm_ev(a->att, w.o2, a->h, nullptr, M, C, C, C, C, w.bo2, a->h, PFPP_ACT_NONE, prec, a, stream));
    if (fuse_ln) {
      TL_CALL(pfpp_layernorm_linear_small(a->h, nullptr, 0, w.g3, w.b3, nullptr, 1, &w.ff1, w.bff1, nullptr, 0, &a->u, inner, M, 2 * inner, C,
                                          eps, stream));
    } else {
      TL_CALL(pfpp_layernorm_split(a->h, a->norm.hi, a->norm.lo, nullptr, 0, w.g3, w.b3, M, C, 1, eps, stream));
      TL_CALL(gemm_ev(a->norm, w.ff1, nullptr, &a->u, M, 2 * inner, C, C, inner, w.bff1, nullptr, PFPP_ACT_GEGLU, prec, a, stream));
    }
    if (small && inner % 512 == 0) TL_CALL(pfpp_gemm_small(&a->u, inner, &w.ff2, w.bff2, a->h, C, a->h, C, M, C, inner, stream));
    else if (wd) TL_CALL(gemm_wd(&a->u, inner, &w.ff2, w.bff2, a->h, C, a->h, C, M, C, inner, stream));
    else TL_CALL(gemm_ev(a->u, w.ff2, a->h, nullptr, M, C, inner, inner, C, w.bff2, a->h, PFPP_ACT_NONE, prec, a, stream));
  }
  return PFPP_OK;
}
