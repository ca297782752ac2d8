"""Training step of the DenoiserTransformer on the HIP kernels.

Reference: Denoiser.forward / _loss / training_step / configure_optimizers (denoiser/model/denoiser.py:80-145,
230-241) around DenoiserTransformer.forward (denoiser_transformer.py:169-203) in train mode, i.e. with the
dropouts of PositionalEncoding (utils/model_utils.py:18-21) and of diffusers' Attention / FeedForward
(attention.py:46-72) active.  The encoder is frozen (train_denoiser.py:33-35): gradients stop at the tokens.

Design
* Only valid fragments are evaluated ("compact" layout, see denoiser.denoiser_forward_compact).  For
  training this is exact, not an approximation: the loss reads valid non-reference fragments only
  (denoiser.py:118-126), no valid token depends on a padded one, so the padded rows of the reference
  contribute exactly zero to every gradient.
* Parameters live in ONE flat fp32 buffer (FlatParams) whose order makes the kernel-side packings free
  views (q|k|v weights adjacent -> [3C, C]; the 12 AdaLN tables / linears adjacent -> batched GEMM operands);
  gradients, Adam moments and the split-f16 planes of the forward GEMMs mirror that buffer, so the optimizer
  is one fused launch and a data-parallel gradient exchange is a handful of large contiguous all-reduces.
  The nn.Parameters of the module are re-pointed at views of the buffer: state_dict keys/shapes are unchanged.
* Backward GEMMs read dY, X and W in place through the k-major loaders of csrc/gemm_grad.hip (no transposed
  copies); gradient operands are lifted by a power-of-two `grad_scale` before the f16 split.
* Dropout masks are regenerated from (seed, site) in the backward, never stored.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from . import train_ops as T
from .packing import PW, round_up

TOKEN_DROPOUT = 0.1          # PositionalEncoding(d_model, dropout=0.1), utils/model_utils.py:13-16


def _param_order(num_layers: int) -> List[str]:
    """flat layout: groups that the kernels read as one operand are adjacent"""
    names: List[str] = []
    lay = [f"transformer_layers.{i}" for i in range(num_layers)]
    for p in lay:
        names += [f"{p}.norm1.emb.weight", f"{p}.norm2.emb.weight"]
    for p in lay:
        names += [f"{p}.norm1.linear.weight", f"{p}.norm2.linear.weight"]
    for p in lay:
        names += [f"{p}.norm1.linear.bias", f"{p}.norm2.linear.bias"]
    for p in lay:
        for a in ("self_attn", "global_attn"):
            names += [f"{p}.{a}.to_q.weight", f"{p}.{a}.to_k.weight", f"{p}.{a}.to_v.weight",
                      f"{p}.{a}.to_out.0.weight", f"{p}.{a}.to_out.0.bias"]
        names += [f"{p}.norm3.weight", f"{p}.norm3.bias", f"{p}.ff.net.0.proj.weight", f"{p}.ff.net.0.proj.bias",
                  f"{p}.ff.net.2.weight", f"{p}.ff.net.2.bias"]
    names += ["shape_embedding.weight", "shape_embedding.bias", "param_fc.weight", "param_fc.bias", "ref_part_emb.weight"]
    for h in ("mlp_out_trans", "mlp_out_rot"):
        for j in (0, 2, 4):
            names += [f"{h}.{j}.weight", f"{h}.{j}.bias"]
    return names


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """t as contiguous fp32 without a launch when it already is"""
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _u8(t: torch.Tensor) -> torch.Tensor:
    """a flag tensor as uint8 without a launch when its storage already is one byte per element (bool)"""
    if t.dtype == torch.uint8:
        return t.contiguous()
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    return (t != 0).to(torch.uint8).contiguous()


def _pw_view(f32: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor) -> PW:
    w = PW.__new__(PW)
    w.scale = 1.0
    w.f32, w.hi, w.lo = f32, hi, lo
    w.N, w.K = int(f32.shape[-2]), int(f32.shape[-1])
    return w


class FlatParams:
    """flat storage of a DenoiserTransformer's parameters, gradients, Adam moments and split-f16 planes"""

    def __init__(self, module: torch.nn.Module):
        named = dict(module.named_parameters())
        self.module = module
        self.num_layers = module.num_layers
        self.order = _param_order(self.num_layers)
        if set(self.order) != set(named):
            raise ValueError(f"FlatParams: unexpected parameter set: {sorted(set(named) ^ set(self.order))}")
        dev = next(module.parameters()).device
        if dev.type != "cuda":
            raise ValueError("FlatParams: the module must live on the GPU (there is no CPU training path)")
        self.offset: Dict[str, int] = {}
        total = 0
        for n in self.order:
            self.offset[n] = total
            total += round_up(named[n].numel(), 8)          # 16-byte aligned fp16 planes
        self.numel = total
        self.params = torch.zeros(total, dtype=torch.float32, device=dev, requires_grad=False)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.hi = torch.zeros(total, dtype=torch.float16, device=dev)
        self.lo = torch.zeros(total, dtype=torch.float16, device=dev)
        self.named = named
        self._odd_buf = None
        self._embed_buf = None
        self.embed_fused = False           # set by the engine: the forward takes the one-launch token embedding
        self._clean = False
        with torch.no_grad():
            for n in self.order:
                p = named[n]
                v = self.view(self.params, n, p.shape)
                v.copy_(p.detach())
                p.data = v
        self.attach_grads()
        self._views: Optional[Dict[str, object]] = None
        self.refresh_planes()
        # layer-wise slices (DDP buckets, in the order the backward completes them)
        self.layer_ranges: List[Tuple[int, int]] = []
        for i in range(self.num_layers):
            a = self.offset[f"transformer_layers.{i}.self_attn.to_q.weight"]
            last = f"transformer_layers.{i}.ff.net.2.bias"
            b = self.offset[last] + round_up(named[last].numel(), 8)
            self.layer_ranges.append((a, b))

    # -- views -------------------------------------------------------------------------------------
    def view(self, flat: torch.Tensor, name: str, shape=None) -> torch.Tensor:
        shape = self.named[name].shape if shape is None else shape
        n = int(math.prod(shape))
        return flat[self.offset[name]: self.offset[name] + n].view(shape)

    def span(self, flat: torch.Tensor, first: str, shape) -> torch.Tensor:
        """a multi-parameter group starting at `first` read as one tensor of `shape`"""
        n = int(math.prod(shape))
        return flat[self.offset[first]: self.offset[first] + n].view(shape)

    def attach_grads(self) -> None:
        """(re-)point every .grad at its slice of the flat gradient buffer.  A parameter whose .grad is None was cleared by
        someone else (module.zero_grad(), a foreign optimizer's zero_grad(set_to_none=True)): its slice still holds the previous
        iteration's gradient and is zeroed here — the backward accumulates, and a silently doubled gradient is the alternative."""
        views = getattr(self, "_grad_views", None)
        if views is None:                  # (parameter, its view of the flat gradient buffer), built once
            views = self._grad_views = [(self.named[n], self.view(self.grads, n)) for n in self.order]
        if all(p.grad is g for p, g in views):          # the common case: nothing touched them since the last call
            return
        dropped = [g for p, g in views if p.grad is None]
        if len(dropped) == len(views):
            self.grads.zero_()
        else:
            for g in dropped:
                g.zero_()
        for p, g in views:
            if p.grad is not g:
                p.grad = g

    def refresh_planes(self) -> None:
        """split-f16 planes of the whole buffer (the optimizer kernel keeps them current afterwards)"""
        with torch.no_grad():
            self.hi.copy_(self.params.to(torch.float16))
            self.lo.copy_((self.params - self.hi.float()).to(torch.float16))
        self._odd = None
        self._seen_version = self._versions()

    def _versions(self):
        """in-place edits of the parameters through torch (load_state_dict, another optimizer) bump these"""
        return tuple(self.named[n]._version for n in self.order)

    def zero_grad(self) -> None:
        if not self._clean:                # optimizer_step(zero_grad=True) already cleared the buffer in its own pass
            self.grads.zero_()
        self._clean = False                # whoever asks for zeros is about to accumulate into them

    # -- kernel-side operands ------------------------------------------------------------------------
    def operands(self) -> Dict[str, object]:
        """weights as the kernels read them: views of the flat buffers (built once), plus the two
        K-padded embeddings (148 / 147 input features) which are re-packed after every optimizer step"""
        first = self.named[self.order[0]]
        if first.data_ptr() != self.params.data_ptr():
            raise RuntimeError("FlatParams: the module's parameters were re-allocated (.to()/.cuda() after the training "
                               "engine was created); build the engine after moving the module")
        if self._versions() != self._seen_version:
            self.refresh_planes()         # load_state_dict / in-place edits of the parameters through torch
        if self._views is None:
            L = self.num_layers
            C = self.named["shape_embedding.bias"].numel()
            n_emb = self.named["transformer_layers.0.norm1.emb.weight"].shape[0]
            v: Dict[str, object] = {}
            t0 = "transformer_layers.0"
            v["ada.tables"] = self.span(self.params, f"{t0}.norm1.emb.weight", (2 * L, n_emb, C))
            v["ada.w"] = _pw_view(*(self.span(f, f"{t0}.norm1.linear.weight", (2 * L, 2 * C, C)) for f in (self.params, self.hi, self.lo)))
            v["ada.b"] = self.span(self.params, f"{t0}.norm1.linear.bias", (2 * L, 2 * C))
            g: Dict[str, torch.Tensor] = {}
            g["ada.tables"] = self.span(self.grads, f"{t0}.norm1.emb.weight", (2 * L, n_emb, C))
            g["ada.w"] = self.span(self.grads, f"{t0}.norm1.linear.weight", (2 * L, 2 * C, C))
            g["ada.b"] = self.span(self.grads, f"{t0}.norm1.linear.bias", (2 * L, 2 * C))

            def lin(key, wname, bname=None, shape=None):
                v[key + ".w"] = _pw_view(*(self.span(f, wname, shape or self.named[wname].shape) for f in (self.params, self.hi, self.lo)))
                g[key + ".w"] = self.span(self.grads, wname, shape or self.named[wname].shape)
                if bname is not None:
                    v[key + ".b"] = self.view(self.params, bname)
                    g[key + ".b"] = self.view(self.grads, bname)

            for i in range(L):
                p = f"transformer_layers.{i}"
                for a in ("self_attn", "global_attn"):
                    lin(f"{i}.{a}.qkv", f"{p}.{a}.to_q.weight", None, (3 * C, C))
                    lin(f"{i}.{a}.o", f"{p}.{a}.to_out.0.weight", f"{p}.{a}.to_out.0.bias")
                v[f"{i}.norm3.g"] = self.view(self.params, f"{p}.norm3.weight")
                v[f"{i}.norm3.b"] = self.view(self.params, f"{p}.norm3.bias")
                g[f"{i}.norm3.g"] = self.view(self.grads, f"{p}.norm3.weight")
                g[f"{i}.norm3.b"] = self.view(self.grads, f"{p}.norm3.bias")
                lin(f"{i}.ff1", f"{p}.ff.net.0.proj.weight", f"{p}.ff.net.0.proj.bias")
                lin(f"{i}.ff2", f"{p}.ff.net.2.weight", f"{p}.ff.net.2.bias")
            for h in ("mlp_out_trans", "mlp_out_rot"):
                for j in (0, 2, 4):
                    lin(f"{h}.{j}", f"{h}.{j}.weight", f"{h}.{j}.bias")
            for key, name in (("shape", "shape_embedding"), ("param", "param_fc")):
                v[key + ".b"] = self.view(self.params, f"{name}.bias")
                g[key + ".b"] = self.view(self.grads, f"{name}.bias")
                g[key + ".w"] = self.view(self.grads, f"{name}.weight")
            v["ref_emb"] = self.view(self.params, "ref_part_emb.weight")
            g["ref_emb"] = self.view(self.grads, "ref_part_emb.weight")
            v["pe"] = self.module.pos_encoding.pe[0].contiguous()
            self._views = {"w": v, "g": g}
        if self._odd is None:
            # the two embeddings with 148 / 147 input features: K-padded copies (fp32 to 4, planes to 8 columns), re-packed after every
            # optimizer step into persistent buffers — one strided copy + one split launch each instead of a dozen torch kernels
            from . import planes as P_

            with torch.no_grad():
                if self.embed_fused:
                    # the one-launch token embedding takes [W_shape | W_param | 0] fragment-blocked, packed from the fp32 parameters in one
                    # launch per optimizer step (csrc/embed_train.hip) instead of the two K-padded copies + splits
                    if self._embed_buf is None:
                        Cw = self.view(self.params, "shape_embedding.bias").numel()
                        dev_ = self.params.device
                        self._embed_buf = (torch.empty(Cw * 320, dtype=torch.float16, device=dev_), torch.empty(Cw * 320, dtype=torch.float16, device=dev_),
                                           torch.empty(Cw, dtype=torch.float32, device=dev_))
                    from . import train_ops as T_

                    T_.embed_pack_weights(self.view(self.params, "shape_embedding.weight"), self.view(self.params, "param_fc.weight"),
                                          self.view(self.params, "shape_embedding.bias"), self.view(self.params, "param_fc.bias"), *self._embed_buf)
                    self._odd = {"embed.w": self._embed_buf[:2], "embed.b": self._embed_buf[2]}
                    w = dict(self._views["w"])
                    w.update(self._odd)
                    return {"w": w, "g": self._views["g"]}
                if self._odd_buf is None:
                    self._odd_buf = {}
                    for key, name in (("shape.w", "shape_embedding.weight"), ("param.w", "param_fc.weight")):
                        wv = self.view(self.params, name)
                        n_, k_ = wv.shape
                        pad8 = torch.zeros((n_, round_up(k_, 8)), dtype=torch.float32, device=wv.device)
                        pl_ = P_.Planes.empty(n_, round_up(k_, 8), wv.device)
                        self._odd_buf[key] = (name, pad8, pl_)
                odd = {}
                for key, (name, pad8, pl_) in self._odd_buf.items():
                    wv = self.view(self.params, name)
                    pad8[:, : wv.shape[1]].copy_(wv)
                    P_.split(pad8, 1.0, out=pl_)
                    pw = _pw_view(pad8, pl_.hi, pl_.lo)          # fp32 rows padded to 8 as well (any multiple of 4 serves the fp32 path)
                    pw.K = int(wv.shape[1])
                    odd[key] = pw
                self._odd = odd
        w = dict(self._views["w"])
        w.update(self._odd)
        return {"w": w, "g": self._views["g"]}

    def after_optimizer_step(self) -> None:
        self._odd = None


class TrainContext:
    """everything the backward needs from one forward"""

    def __init__(self):
        self.t: Dict[str, object] = {}
        self._release = None          # hands the forward arena back to the engine's pool (end of the backward, or when dropped)

    def release(self) -> None:
        r, self._release = self._release, None
        if r is not None:
            r()

    def __del__(self):
        try:
            self.release()
        except Exception:             # interpreter shutdown
            pass


_DIAG_HOST_DELAY_US = float(os.environ.get("PFPP_DIAG_HOST_DELAY_US", "0"))


def _traced_wd() -> bool:
    """bench.py's per-launch timing pass (ops.GEMM_TRACE) issues the blocks from Python; it then takes the weight-direct GEMMs the C
    sequencer takes (PFPP_TRAIN_WD), so that the roofline object describes the kernels of the timed region — the untraced Python
    sequence stays the tiled cross-check"""
    return ops.GEMM_TRACE is not None and os.environ.get("PFPP_TRAIN_WD", "1") == "1" and not ops.SINGLE_PASS


class DenoiserTrainEngine:
    """forward (train mode) / backward / optimizer step of a DenoiserTransformer on the HIP kernels"""

    def __init__(self, module: torch.nn.Module, *, dropout: Optional[float] = None, token_dropout: float = TOKEN_DROPOUT,
                 grad_scale: float = 4096.0):
        self.flat = FlatParams(module)
        # token embedding: forward in one launch on the packed [W_shape | W_param] planes, backward in one launch on the transposed feature
        # planes (csrc/embed_train.hip); 0 = features + two skinny GEMMs + combine / two weight-gradient GEMMs + sums (cross-checks)
        shp = getattr(module, "shape_embedding", None), getattr(module, "param_fc", None)
        odd_ok = all(m_ is not None for m_ in shp) and shp[0].weight.shape[1] == 148 and shp[1].weight.shape[1] == 147 and shp[0].weight.shape[0] % 32 == 0
        self._embed_bwd_fused = odd_ok and os.environ.get("PFPP_TRAIN_EMBED_BWD_FUSED", "1") != "0"
        self._embed_fwd_fused = odd_ok and os.environ.get("PFPP_TRAIN_EMBED_FWD_FUSED", "1") != "0"
        self.flat.embed_fused = self._embed_fwd_fused
        self.module = module
        self.num_layers = module.num_layers
        self.num_heads = module.num_heads
        self.p_layer = 0.2 if dropout is None else float(dropout)    # EncoderLayer(dropout=0.2), denoiser_transformer.py
        self.p_token = float(token_dropout)
        if math.log2(grad_scale) % 1 != 0:
            raise ValueError("grad_scale must be a power of two (exact rescaling)")
        self.grad_scale = float(grad_scale)
        # dynamic gradient scale: grad_scale follows the magnitude of the loss gradient (a power of two keeping max |dpred| * G
        # in [8, 16), which is where the default 4096 puts an untrained model) so that the f16 split of the backward operands
        # keeps its precision when the loss — and with it every gradient — shrinks during training.  The magnitude is the one
        # observed two backward passes ago (read from pinned memory: no stall, no device read in the step, deterministic).
        self._dyn_gscale = os.environ.get("PFPP_TRAIN_DYN_GSCALE", "1") != "0"
        self._ada_per_layer = os.environ.get("PFPP_TRAIN_ADA_PER_LAYER", "1") != "0"
        self._ada_bwd_fused = os.environ.get("PFPP_TRAIN_ADA_BWD_FUSED", "1") != "0"    # 0 = column sum + two tiled gradient GEMMs (cross-check)
        self._ada_layerwise = None
        self._amax_ring = None
        self._n_backward = 0
        self.step_count = 0
        from .parallel import GradExchange

        # the 12 AdaLN timestep tables open the flat buffer (see _param_order): exchanged as rows, not as 75 MB of zeros
        self._sparse_tables = os.environ.get("PFPP_SPARSE_TABLE_GRADS", "1") == "1"
        n_tab = self.flat.offset[f"transformer_layers.0.norm1.linear.weight"]       # the tables are the first group of the layout
        self._exchange = GradExchange(self.flat.grads, self.flat.layer_ranges, (0, n_tab) if self._sparse_tables else (0, 0))
        # overflow guard (the GradScaler of this engine): the backward's operands are fp16 planes of grad_scale * dY, written without
        # saturation — a batch whose gradients outgrow the lagged scale estimate yields inf / NaN gradients.  The AdamW launches
        # skip and flag such elements on the device (pfpp_adamw_guarded: parameters and moments are never poisoned, no host read);
        # the flag travels to the host through pinned memory two steps later and backs the scale off (x 1/16, regrown x 2 per
        # 200 clean steps).  PFPP_TRAIN_OVERFLOW_GUARD=0 removes the guard.
        self._guard = os.environ.get("PFPP_TRAIN_OVERFLOW_GUARD", "1") != "0"
        self._overflow = torch.zeros(2, dtype=torch.int32, device=self.flat.params.device) if self._guard else None
        self._ovf_ring = None
        self._backoff = 1.0                          # power of two <= 1 applied on top of the dpred-tracking scale
        self._clean_steps = 0
        self.overflow_steps = 0                      # optimizer steps in which non-finite gradients were seen (and skipped)
        # weight / bias gradients are off the critical path (only the optimizer needs them): they run on a second
        # HIP stream next to the dX chain, which by itself launches too few workgroups to fill 256 CUs
        self._side = ((_masked_stream(self.flat.params.device, int(os.environ.get("PFPP_SIDE_CU_FRACTION_PCT", "0")), from_top=True) or
                       torch.cuda.Stream(device=self.flat.params.device, priority=int(os.environ.get("PFPP_SIDE_PRIORITY", "0"))))
                      if os.environ.get("PFPP_TRAIN_SIDE_STREAM", "1") == "1" else None)
        # every dropout site is followed by a LayerNorm (forward) / follows a LayerNorm backward: one launch for both
        self._fuse_drop = os.environ.get("PFPP_TRAIN_FUSE_DROP", "1") != "0"
        self._fuse_colsum = os.environ.get("PFPP_TRAIN_FUSE_COLSUM", "1") != "0"   # bias gradients from the weight-gradient GEMM's dY tiles
        # plane path (default): every transformer-block GEMM of the step — forward, dX = dY.W and dW = dY^T.X — runs on the
        # LDS-DMA staged plane kernel (csrc/gemm_pl.hip); the LayerNorm / attention / GEGLU kernels hand their results over
        # as split-f16 planes, gradients lifted by grad_scale.  PFPP_TRAIN_PLANES=0 restores the register-staged kernels.
        self._planes = os.environ.get("PFPP_TRAIN_PLANES", "1") == "1" and ops.GEMM_MODE == "f16x3"
        # the six blocks' ~240 launches per iteration enqueued from C (csrc/tlayer.hip: pfpp_tlayers_fwd / _bwd) instead of one ctypes
        # call each; same launches, same arguments (PFPP_TRAIN_CSEQ=0: the Python sequence below, the cross-check of the tests)
        self._cseq = os.environ.get("PFPP_TRAIN_CSEQ", "1") == "1"
        # pool -> both output heads as one launch, their backward as one launch + one grouped weight-gradient launch (csrc/heads.hip);
        # 0 = the layer-wise GEMMs / activations (the cross-check of the tests)
        self._heads_fused = os.environ.get("PFPP_HEADS_FUSED", "1") == "1" and ops.GEMM_MODE == "f16x3"
        self._heads_static = None
        self._cseq_static = None
        self._arena_pool: Dict[tuple, list] = {}
        self._tables_rows = None                      # armed single-rank step: (timesteps, table shape, end of the table range) — rows updated early
        self._ada_ranges = []
        self._ada_static = (None, None)
        self._dw_pending = None                       # traced Python backward: the block's weight gradients collected for one grouped launch
        self._armed = None                            # arm_optimizer(): hyper-parameters of an optimizer-in-backward step
        self._armed_zero = False
        self._early: List[Tuple[int, int]] = []       # [a, b) ranges of the flat buffer the armed backward has already updated
        self._sync = True                            # False inside no_sync(): this backward does not start the gradient exchange
        self._exchanged = False                      # the gradients in the flat buffer have been all-reduced since the last step
        self._accumulated = False                    # a no_sync backward has accumulated into the buffer since the last step
        # rows of the AdaLN timestep tables that have ever received a gradient (bitmap, set by silu_embed_bwd): the others have zero
        # moments and — with fl32(1 - lr * weight_decay) = 1, the reference's hyper-parameters — an AdamW update that is exactly the
        # identity, so the iteration's closing update skips them (two thirds of the tables: 12.7 M of 57.6 M parameters).
        # PFPP_TRAIN_TABLES_ACTIVE=0: every row every step (cross-check).  None = not tracked (every row is taken)
        self._join_pending = False                   # the last backward left the join with the weight-gradient stream to optimizer_step()
        self._tail_overlap = os.environ.get("PFPP_TRAIN_TAIL_OVERLAP", "1") != "0"
        self._heads_dw_ev: Optional[torch.cuda.Event] = None
        self._heads_ev_fresh = False                 # ... recorded by the backward that is running
        self._tab_active: Optional[torch.Tensor] = None
        if os.environ.get("PFPP_TRAIN_TABLES_ACTIVE", "1") != "0":
            self._tab_active = torch.zeros(128, dtype=torch.int32, device=self.flat.params.device)
            self.tables_state_changed()

    def tables_state_changed(self) -> None:
        """the optimizer moments of the timestep tables were written from outside (load_state_dict, a restored checkpoint): rebuild the
        bitmap of rows whose update is not the identity = rows with a non-zero moment or gradient in any table"""
        if self._tab_active is None:
            return
        f = self.flat
        name = "transformer_layers.0.norm1.emb.weight"
        if name not in f.offset:
            self._tab_active = None
            return
        n_emb = f.named[name].shape[0]
        n_tab = f.offset["transformer_layers.0.norm1.linear.weight"]
        C_ = f.named[name].shape[1]
        if n_emb > 4096 or f.offset[name] != 0 or n_tab % (n_emb * C_):
            self._tab_active = None
            return
        live = torch.zeros(n_emb, dtype=torch.bool, device=f.params.device)
        for buf in (f.exp_avg, f.exp_avg_sq, f.grads):
            live |= (buf[:n_tab].view(-1, n_emb, C_) != 0).any(dim=2).any(dim=0)
        bits = torch.zeros(128 * 32, dtype=torch.int64, device=f.params.device)
        bits[:n_emb] = live.to(torch.int64)
        words = (bits.view(128, 32) << torch.arange(32, device=bits.device)).sum(dim=1)
        self._tab_active.copy_(torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32))

    def single_stream(self) -> None:
        """everything on the caller's stream from now on (profiling / per-kernel timing)"""
        self._side = None

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, timesteps, latent, xyz, part_valids, scale, ref_part, *, seed: int = 0,
                train: bool = True) -> Tuple[torch.Tensor, TrainContext]:
        """-> (pred_noise [B,P,7] with zeros at padded slots, context).  `train=False` disables the dropouts
        (the reference module in .eval())."""
        self._join_side()                 # (a backward whose optimizer_step() was skipped)
        B, P, L, _ = latent.shape
        # the one-launch token embedding (forward and backward) exists for >= 11 latent points of width 64 in the split-f16 mode
        # (pfpp_embed_tokens_small returns PFPP_EUNSUPPORTED otherwise): decided per call, like denoiser.py does for the eval path;
        # anything else takes features + two linears + combine / the two weight-gradient GEMMs
        embed_ok = L >= 11 and latent.shape[-1] == 64 and ops.GEMM_MODE == "f16x3" and not ops.SINGLE_PASS
        if self.flat.embed_fused != (self._embed_fwd_fused and embed_ok):
            self.flat.embed_fused = self._embed_fwd_fused and embed_ok
            self.flat._odd = None                     # re-pack the two odd-width embeddings in the layout this call takes
        ops_ = self.flat.operands()
        w = ops_["w"]
        C = w["shape.b"].numel()
        H = self.num_heads
        dh = C // H
        n_slots = B * P
        dev = latent.device
        p_tok = self.p_token if train else 0.0
        p_lay = self.p_layer if train else 0.0
        ctx = TrainContext()
        s = ctx.t
        from .denoiser import layout_of

        lay = layout_of(part_valids, L)                     # remembered on the tensor: no device read for a repeated batch
        slot, Fv, frag_b, frag_p = lay.slot, lay.Fv, lay.frag_b, lay.frag_p
        if Fv == 0:
            raise ValueError("training forward: the batch has no valid fragment")
        seq_len, seq_off, max_len = lay.seq_len, lay.seq_off, lay.max_len
        M = Fv * L
        # the valid-fragment gather of the inputs happens inside the kernels (slot32): no gathered copies of latent / xyz / scale / x / ref
        slot32 = lay.slot32
        raw = (_f32c(latent).reshape(n_slots, L, -1), _f32c(xyz).reshape(n_slots, L, 3), _f32c(scale).reshape(n_slots), _f32c(x).reshape(n_slots, 7))
        ref_u8 = _u8(ref_part).reshape(n_slots)           # padded flags; the kernels read ref_u8[slot[f]]
        sf = pf = ft = None
        if self._embed_bwd_fused and embed_ok:
            # the backward's operand: the extended feature rows transposed as split-f16 planes (csrc/embed_train.hip)
            ft = T.token_features_t(*raw, slot32, ref_u8, Fv, L)
        if "embed.w" in w:
            h = T.embed_tokens_packed(*raw, slot32, *w["embed.w"], w["embed.b"], w["ref_emb"], ref_u8, w["pe"], frag_p, Fv, L)
        if "embed.w" not in w or ft is None:
            sf, pf = ops.token_features(*raw, slot=slot32)
        if "embed.w" not in w:
            shape_emb = ops.linear(sf, w["shape.w"], w["shape.b"])
            x_emb = ops.linear(pf, w["param.w"], w["param.b"])
            h = ops.token_combine_list(shape_emb, x_emb, w["ref_emb"], ref_u8, w["pe"], frag_p, L, slot=slot32)
        fuse = self._fuse_drop                        # dropout sites ride in the LayerNorm kernels that follow them
        if p_tok > 0.0 and not fuse:
            T.dropout(h, p_tok, seed, 0, out=h)
        n_ada = 2 * self.num_layers
        t64 = timesteps.to(torch.int64).contiguous()
        se = ops.silu_embed(w["ada.tables"], t64)
        mods = torch.empty((n_ada, B, 2 * C), dtype=torch.float32, device=dev)
        ops.gemm(se, w["ada.w"], M=B, N=2 * C, K=C, lda=C, out=mods, ldc=2 * C, bias=w["ada.b"],
                 batch=n_ada, sA=(B * C, 0), sW=(2 * C * C, 0), sC=(B * 2 * C, 0), sV=(2 * C, 0))
        att_scale = 1.0 / math.sqrt(dh)
        s.update(dict(B=B, P=P, L=L, C=C, Fv=Fv, M=M, slot=slot, frag_b=frag_b, seq_len=seq_len, seq_off=seq_off,
                      max_len=max_len, sf=sf, pf=pf, ref_u8=ref_u8, slot32=slot32, t64=t64, se=se, mods=mods, seed=seed, p_tok=p_tok,
                      p_lay=p_lay, att_scale=att_scale, n_slots=n_slots, fuse=fuse, ft=ft))
        layers = []
        if self._planes and self._use_cseq(fuse):
            h = self._forward_layers_c(h, w, s, mods, p_tok, p_lay, seed, Fv, L, H, att_scale)
            ctx._release = lambda ka=s.pop("_fwd_arena"): self._give_arena(*ka)
        elif self._planes:
            h = self._forward_layers_planes(h, w, s, mods, layers, p_tok, p_lay, seed, fuse, Fv, L, H, dh, att_scale)
        else:
            h = self._forward_layers(h, w, s, mods, layers, p_tok, p_lay, seed, fuse, Fv, L, H, dh, att_scale, M, C)
        s["layers"] = layers
        s["dx_pool"] = None if self._planes else self._zeroed_dx_pool(M, C, w)
        pooled = ops.mean_pool(h, Fv, L)
        s["pooled"] = pooled
        out = torch.zeros((n_slots, 7), dtype=torch.float32, device=dev)
        if self._heads_fused and C == 512:
            trans, rot, _, _ = self._head_structs(w, ops_["g"])
            s["heads_saved"] = T.heads_fwd(pooled, trans, rot, out, slot=lay.slot32, save=True)
            return out.view(B, P, 7), ctx
        out_c = torch.empty((Fv, 7), dtype=torch.float32, device=dev)
        heads = {}
        for name, c0, width in (("mlp_out_trans", 0, 3), ("mlp_out_rot", 3, 4)):
            a0 = ops.linear(pooled, w[f"{name}.0.w"], w[f"{name}.0.b"])
            v0 = T.act(a0, "silu")
            a1 = ops.linear(v0, w[f"{name}.2.w"], w[f"{name}.2.b"])
            v1 = T.act(a1, "silu")
            ops.gemm(v1, w[f"{name}.4.w"], M=Fv, N=width, K=v1.shape[1], lda=v1.shape[1], out=out_c, ldc=7,
                     bias=w[f"{name}.4.b"], c_off=c0)
            heads[name] = (a0, v0, a1, v1)
        s["heads"] = heads
        ops.scatter_rows(out_c, slot.to(torch.int32).contiguous(), n_slots, out=out)
        return out.view(B, P, 7), ctx

    def _forward_layers(self, h, w, s, mods, layers, p_tok, p_lay, seed, fuse, Fv, L, H, dh, att_scale, M, C):
        """transformer blocks on the register-staged GEMMs (fp32 activations between kernels)"""
        frag_b, seq_off, seq_len, max_len = s["frag_b"], s["seq_off"], s["seq_len"], s["max_len"]
        for i in range(self.num_layers):
            lay: Dict[str, torch.Tensor] = {}
            if i == 0 and fuse and p_tok > 0.0:
                h, lay["n1"] = T.dropout_layernorm(h, None, p_tok, seed, 0, mod=mods[0], group_batch=frag_b, group_rows=L)
            else:
                lay["n1"] = ops.layernorm_grouped(h, mods[2 * i], frag_b, L)
            lay["h0"] = h
            lay["qkv1"] = ops.linear(lay["n1"], w[f"{i}.self_attn.qkv.w"])
            lay["att1"] = ops.attn_blockdiag(lay["qkv1"], Fv, L, H, dh, att_scale)
            if fuse and p_lay > 0.0:
                y = ops.gemm(lay["att1"], w[f"{i}.self_attn.o.w"], M=M, N=C, K=C, lda=C, ldc=C, bias=w[f"{i}.self_attn.o.b"])
                h, lay["n2"] = T.dropout_layernorm(y, h, p_lay, seed, 1 + 3 * i, mod=mods[2 * i + 1], group_batch=frag_b, group_rows=L)
            else:
                h = self._proj_residual(lay["att1"], w[f"{i}.self_attn.o.w"], w[f"{i}.self_attn.o.b"], h, p_lay, seed, 1 + 3 * i)
                lay["n2"] = ops.layernorm_grouped(h, mods[2 * i + 1], frag_b, L)
            lay["h1"] = h
            lay["qkv2"] = ops.linear(lay["n2"], w[f"{i}.global_attn.qkv.w"])
            lay["att2"], lay["lse"] = T.attn_dense_train(lay["qkv2"], seq_off, seq_len, max_len, H, dh, att_scale)
            if fuse and p_lay > 0.0:
                y = ops.gemm(lay["att2"], w[f"{i}.global_attn.o.w"], M=M, N=C, K=C, lda=C, ldc=C, bias=w[f"{i}.global_attn.o.b"])
                h, lay["n3"] = T.dropout_layernorm(y, h, p_lay, seed, 2 + 3 * i, gamma=w[f"{i}.norm3.g"], beta=w[f"{i}.norm3.b"])
            else:
                h = self._proj_residual(lay["att2"], w[f"{i}.global_attn.o.w"], w[f"{i}.global_attn.o.b"], h, p_lay, seed, 2 + 3 * i)
                lay["n3"] = ops.layernorm(h, gamma=w[f"{i}.norm3.g"], beta=w[f"{i}.norm3.b"])
            lay["h2"] = h
            lay["z"] = ops.linear(lay["n3"], w[f"{i}.ff1.w"], w[f"{i}.ff1.b"])
            lay["u"] = T.geglu(lay["z"], p_lay, seed, 3 + 3 * i)
            inner = lay["u"].shape[1]
            h = ops.gemm(lay["u"], w[f"{i}.ff2.w"], M=M, N=C, K=inner, lda=inner, ldc=C, bias=w[f"{i}.ff2.b"], residual=h, ldr=C)
            layers.append(lay)
        return h

    def _forward_layers_planes(self, h, w, s, mods, layers, p_tok, p_lay, seed, fuse, Fv, L, H, dh, att_scale):
        """transformer blocks on the plane GEMM: every GEMM operand our own kernels produce (normalised rows, attention
        outputs, GEGLU products) is written as split-f16 planes by its producer and kept for the backward's dW = dY^T.X"""
        from . import planes as P

        frag_b, seq_off, seq_len, max_len = s["frag_b"], s["seq_off"], s["seq_len"], s["max_len"]
        M, C = h.shape
        dev = h.device

        def wp(key):
            pw = w[key]
            return P.Planes(pw.hi, pw.lo, pw.scale)

        # traced pass: every layer's five weights blocked by ONE launch up front, as pfpp_tlayers_fwd does for a six-layer call
        frags = self._traced_frags(w, transposed=False) if _traced_wd() else {}

        def lin(a, key, N, K, bias=None, residual=None):
            out = torch.empty((M, N), dtype=torch.float32, device=dev)
            if _traced_wd() and not key.endswith("ff1.w") and N % 128 == 0 and K % 64 == 0:
                return P.gemm_wd(a, wp(key), out, M=M, N=N, K=K, bias=bias, residual=residual, frag=frags.get(key))
            return P.gemm(a, wp(key), out, M=M, N=N, K=K, bias=bias, residual=residual)

        def ln_planes(x, i_mod=None, gamma=None, beta=None):
            n = ops.SplitAct.empty(M, C, dev)
            if i_mod is not None:
                ops.layernorm_grouped(x, mods[i_mod], frag_b, L, out=n)
            else:
                ops.layernorm(x, gamma=gamma, beta=beta, out=n)
            return P.Planes(n.hi, n.lo)

        for i in range(self.num_layers):
            lay: Dict[str, object] = {}
            if i == 0 and fuse and p_tok > 0.0:
                h, lay["n1"] = T.dropout_layernorm_planes(h, None, p_tok, seed, 0, mod=mods[0], group_batch=frag_b, group_rows=L)
            else:
                lay["n1"] = ln_planes(h, i_mod=2 * i)
            lay["h0"] = h
            lay["qkv1"] = lin(lay["n1"], f"{i}.self_attn.qkv.w", 3 * C, C)
            a1 = ops.SplitAct.empty(M, C, dev)
            ops.attn_blockdiag(lay["qkv1"], Fv, L, H, dh, att_scale, out=a1)
            lay["att1"] = P.Planes(a1.hi, a1.lo)
            if fuse and p_lay > 0.0:
                y = lin(lay["att1"], f"{i}.self_attn.o.w", C, C, bias=w[f"{i}.self_attn.o.b"])
                h, lay["n2"] = T.dropout_layernorm_planes(y, h, p_lay, seed, 1 + 3 * i, mod=mods[2 * i + 1], group_batch=frag_b, group_rows=L)
            else:
                if p_lay > 0.0:
                    y = lin(lay["att1"], f"{i}.self_attn.o.w", C, C, bias=w[f"{i}.self_attn.o.b"])
                    h = T.dropout(y, p_lay, seed, 1 + 3 * i, res=h, out=y)
                else:
                    h = lin(lay["att1"], f"{i}.self_attn.o.w", C, C, bias=w[f"{i}.self_attn.o.b"], residual=h)
                lay["n2"] = ln_planes(h, i_mod=2 * i + 1)
            lay["h1"] = h
            lay["qkv2"] = lin(lay["n2"], f"{i}.global_attn.qkv.w", 3 * C, C)
            lay["att2"], lay["att2p"], lay["lse"] = T.attn_dense_train_planes(lay["qkv2"], seq_off, seq_len, max_len, H, dh, att_scale)
            if fuse and p_lay > 0.0:
                y = lin(lay["att2p"], f"{i}.global_attn.o.w", C, C, bias=w[f"{i}.global_attn.o.b"])
                h, lay["n3"] = T.dropout_layernorm_planes(y, h, p_lay, seed, 2 + 3 * i, gamma=w[f"{i}.norm3.g"], beta=w[f"{i}.norm3.b"])
            else:
                if p_lay > 0.0:
                    y = lin(lay["att2p"], f"{i}.global_attn.o.w", C, C, bias=w[f"{i}.global_attn.o.b"])
                    h = T.dropout(y, p_lay, seed, 2 + 3 * i, res=h, out=y)
                else:
                    h = lin(lay["att2p"], f"{i}.global_attn.o.w", C, C, bias=w[f"{i}.global_attn.o.b"], residual=h)
                lay["n3"] = ln_planes(h, gamma=w[f"{i}.norm3.g"], beta=w[f"{i}.norm3.b"])
            lay["h2"] = h
            inner2 = w[f"{i}.ff1.w"].f32.shape[0]
            lay["z"] = lin(lay["n3"], f"{i}.ff1.w", inner2, C, bias=w[f"{i}.ff1.b"])
            lay["u"] = T.geglu_planes(lay["z"], p_lay, seed, 3 + 3 * i)
            h = lin(lay["u"], f"{i}.ff2.w", C, inner2 // 2, bias=w[f"{i}.ff2.b"], residual=h)
            layers.append(lay)
        return h

    # ------------------------------------------------------------------------------------------ blocks sequenced from C
    def _traced_frags(self, w, transposed: bool):
        """fragment-blocked planes of every layer's qkv / out / second feed-forward weights (of their transposes: the input-gradient
        operands) from ONE reblock launch -> {weight key: (fhi, flo)}; only the weights whose widths the blocked layout covers"""
        from . import planes as P

        keys = [f"{i}.{k}" for i in range(self.num_layers) for k in ("self_attn.qkv.w", "self_attn.o.w", "global_attn.qkv.w", "global_attn.o.w", "ff2.w")]
        ok = [k for k in keys if w[k].hi.shape[0] % 64 == 0 and w[k].hi.shape[1] % 64 == 0]
        if not ok or len(ok) > 32:
            return {}
        out = P.reblock_many([(P.Planes(w[k].hi, w[k].lo, w[k].scale), transposed) for k in ok])
        return dict(zip(ok, out))

    def _take_arena(self, kind: str, nbytes: int, dev) -> torch.Tensor:
        """a persistent byte arena of the C-sequenced path.  Reuse needs no event: an arena goes back to the pool at the end of the
        backward that read it — after the main stream was made to wait for the weight-gradient stream (_all_done) — and its next user
        writes it from the same main stream, i.e. behind all of that in stream order.  Keyed by (kind, stream, size): a forward on
        another stream, or a second forward before the first one's backward, simply gets another arena."""
        key = (kind, ops.raw_stream_id(dev.index), nbytes)
        free = self._arena_pool.setdefault(key, [])
        buf = free.pop() if free else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return key, buf

    def _give_arena(self, key, buf) -> None:
        free = self._arena_pool.setdefault(key, [])
        if len(free) < 3:
            free.append(buf)

    def _use_cseq(self, fuse: bool) -> bool:
        return self._cseq and fuse and ops.GEMM_TRACE is None and not ops.SINGLE_PASS

    def _cseq_args(self, w, g):
        """the per-engine part of pfpp_tlayers_args: weight planes / bias / gradient / optimizer-slice pointers into the flat buffers
        (which never move: FlatParams.operands() checks) -> (args, keep-alive)"""
        if self._cseq_static is not None:
            return self._cseq_static
        from ._lib import PlanesC, TlayerAdamw, TlayerGrads, TlayerParams, TlayersArgs

        n = self.num_layers
        f = self.flat
        layers, grads, adam = (TlayerParams * n)(), (TlayerGrads * n)(), (TlayerAdamw * n)()
        for i in range(n):
            for name, key in (("qkv1", f"{i}.self_attn.qkv.w"), ("o1", f"{i}.self_attn.o.w"), ("qkv2", f"{i}.global_attn.qkv.w"),
                              ("o2", f"{i}.global_attn.o.w"), ("ff1", f"{i}.ff1.w"), ("ff2", f"{i}.ff2.w")):
                pw = w[key]
                setattr(layers[i], name, PlanesC(pw.hi.data_ptr(), pw.lo.data_ptr(), pw.scale))
            for name, key in (("bo1", f"{i}.self_attn.o.b"), ("bo2", f"{i}.global_attn.o.b"), ("g3", f"{i}.norm3.g"), ("b3", f"{i}.norm3.b"),
                              ("bff1", f"{i}.ff1.b"), ("bff2", f"{i}.ff2.b")):
                setattr(layers[i], name, w[key].data_ptr())
            for name, key in (("qkv1_w", f"{i}.self_attn.qkv.w"), ("o1_w", f"{i}.self_attn.o.w"), ("o1_b", f"{i}.self_attn.o.b"),
                              ("qkv2_w", f"{i}.global_attn.qkv.w"), ("o2_w", f"{i}.global_attn.o.w"), ("o2_b", f"{i}.global_attn.o.b"),
                              ("g3", f"{i}.norm3.g"), ("b3", f"{i}.norm3.b"), ("ff1_w", f"{i}.ff1.w"), ("ff1_b", f"{i}.ff1.b"),
                              ("ff2_w", f"{i}.ff2.w"), ("ff2_b", f"{i}.ff2.b")):
                setattr(grads[i], name, g[key].data_ptr())
            a, b = f.layer_ranges[i]
            adam[i] = TlayerAdamw(f.params[a:b].data_ptr(), f.grads[a:b].data_ptr(), f.exp_avg[a:b].data_ptr(), f.exp_avg_sq[a:b].data_ptr(),
                                  f.hi[a:b].data_ptr(), f.lo[a:b].data_ptr(), b - a)
        # the blocks' AdaLN linears (stacked [2 n, 2C, C] / [2 n, 2C]): block i's slices [2 i, 2 i + 2) for the per-block gradients /
        # AdamW the C sequencer queues on the side stream (single rank)
        ada_w_ad, ada_b_ad = (TlayerAdamw * n)(), (TlayerAdamw * n)()
        t0 = "transformer_layers.0"
        aw, ab = f.offset[f"{t0}.norm1.linear.weight"], f.offset[f"{t0}.norm1.linear.bias"]
        Cw = w["shape.b"].numel()
        for i in range(n):
            x0, x1 = aw + 2 * i * 2 * Cw * Cw, aw + (2 * i + 2) * 2 * Cw * Cw
            y0, y1 = ab + 2 * i * 2 * Cw, ab + (2 * i + 2) * 2 * Cw
            ada_w_ad[i] = TlayerAdamw(f.params[x0:x1].data_ptr(), f.grads[x0:x1].data_ptr(), f.exp_avg[x0:x1].data_ptr(),
                                      f.exp_avg_sq[x0:x1].data_ptr(), f.hi[x0:x1].data_ptr(), f.lo[x0:x1].data_ptr(), x1 - x0)
            ada_b_ad[i] = TlayerAdamw(f.params[y0:y1].data_ptr(), f.grads[y0:y1].data_ptr(), f.exp_avg[y0:y1].data_ptr(),
                                      f.exp_avg_sq[y0:y1].data_ptr(), f.hi[y0:y1].data_ptr(), f.lo[y0:y1].data_ptr(), y1 - y0)
        self._ada_ranges = [(aw, aw + 2 * n * 2 * Cw * Cw), (ab, ab + 2 * n * 2 * Cw)]
        self._ada_static = (ada_w_ad, ada_b_ad)
        args = TlayersArgs()
        args.n_layers = n
        args.layers, args.grads = layers, grads
        C = w["shape.b"].numel()
        args.C, args.H, args.inner = C, self.num_heads, w["0.ff1.w"].f32.shape[0] // 2
        if os.environ.get("PFPP_TRAIN_WD", "1") == "1":
            # scratch for the fragment-blocked copies of a layer's weights: qkv / out / second feed-forward linears and their input
            # gradients with the weights read straight into the matrix operands (csrc/gemm_wd.hip; 0 = the tiled kernel, the cross-check)
            from . import _lib
            nbytes = int(_lib.load().pfpp_tlayers_frag_bytes(C, int(args.inner))) * n      # room for every layer: one blocking launch per call
            self._frag_ws = torch.empty(nbytes, dtype=torch.uint8, device=w["shape.b"].device)
            args.frag_ws, args.frag_ws_bytes = self._frag_ws.data_ptr(), nbytes
        self._cseq_static = (args, (layers, grads, adam))
        return self._cseq_static

    def _forward_layers_c(self, h, w, s, mods, p_tok, p_lay, seed, Fv, L, H, att_scale):
        """_forward_layers_planes with the launches enqueued by pfpp_tlayers_fwd; the saved activations live in one arena"""
        import ctypes as C_

        from . import _lib, planes as P

        args, _keep = self._cseq_args(w, self.flat.operands()["g"])
        lib = _lib.load()
        M, C = h.shape
        dev = h.device
        inner = int(args.inner)
        layer_bytes = int(lib.pfpp_tlayers_fwd_bytes(M, C, H, inner))
        key, arena = self._take_arena("fwd", self.num_layers * layer_bytes, dev)
        s["_fwd_arena"] = (key, arena)
        main_h = ops.raw_stream_id(dev.index)
        args.M, args.L, args.Fv, args.B = M, L, Fv, mods.shape[1]
        args.h_in, args.mods = h.data_ptr(), mods.data_ptr()
        args.frag_b, args.seq_off, args.seq_len = s["frag_b"].data_ptr(), s["seq_off"].data_ptr(), s["seq_len"].data_ptr()
        args.n_seq, args.max_len = s["seq_off"].numel(), s["max_len"]
        args.att_scale, args.p_tok, args.p_lay, args.seed = att_scale, p_tok, p_lay, seed
        args.fwd_arena, args.fwd_layer_bytes = arena.data_ptr(), layer_bytes
        ws = P.workspace_for(dev, main_h)
        args.ws_main, args.ws_bytes = ws.data_ptr(), ws.numel() * 4
        _lib.check(lib.pfpp_tlayers_fwd(C_.byref(args), 0, self.num_layers, C_.c_void_p(main_h)), "pfpp_tlayers_fwd")
        hout = int(lib.pfpp_tlayers_fwd_hout_offset(M, C, H, inner)) + (self.num_layers - 1) * layer_bytes
        s["cseq"] = dict(arena=arena, layer_bytes=layer_bytes, tokens=h, mods=mods)
        return arena[hout: hout + M * C * 4].view(torch.float32).view(M, C)

    def _backward_layers_c(self, s, w, g, dh_, dmods):
        """_backward_layers_planes with the launches enqueued by pfpp_tlayers_bwd: one call for all layers (the per-layer AdamW of
        an armed step included), or one call per layer when the ranks exchange each layer's gradients as it completes"""
        import ctypes as C_

        from . import _lib, planes as P
        from ._lib import PlanesC

        args, (layers, grads, adam) = self._cseq_args(w, g)
        lib = _lib.load()
        cs = s["cseq"]
        M, C, L, Fv = s["M"], s["C"], s["L"], s["Fv"]
        H, inner = self.num_heads, int(args.inner)
        dev = dh_.device
        G = self.grad_scale
        main_h = ops.raw_stream_id(dev.index)
        side_h = self._side.cuda_stream if self._side is not None else None
        bwd_bytes = int(lib.pfpp_tlayers_bwd_bytes(M, C, H, inner))
        tmp_key, tmp = self._take_arena("bwd", bwd_bytes, dev)
        s["_bwd_tmp"] = (tmp_key, tmp)
        dhp = P.split(dh_, G)
        p_tok = s["p_tok"]
        dtok = torch.empty_like(dh_) if p_tok > 0.0 else dh_
        if self._side is not None:
            dhp.record_stream(self._side)
        args.M, args.L, args.Fv, args.B = M, L, Fv, cs["mods"].shape[1]
        args.h_in, args.mods = cs["tokens"].data_ptr(), cs["mods"].data_ptr()
        args.frag_b, args.seq_off, args.seq_len = s["frag_b"].data_ptr(), s["seq_off"].data_ptr(), s["seq_len"].data_ptr()
        args.n_seq, args.max_len = s["seq_off"].numel(), s["max_len"]
        args.att_scale, args.p_tok, args.p_lay, args.seed = s["att_scale"], p_tok, s["p_lay"], s["seed"]
        args.fwd_arena, args.fwd_layer_bytes = cs["arena"].data_ptr(), cs["layer_bytes"]
        ws = P.workspace_for(dev, main_h)
        args.ws_main, args.ws_bytes = ws.data_ptr(), ws.numel() * 4
        args.ws_side = P.workspace_for(dev, side_h).data_ptr() if side_h is not None else None
        args.bwd_arena, args.bwd_bytes = tmp.data_ptr(), bwd_bytes
        args.grad_scale = G
        args.dh, args.dmods, args.dtok = dh_.data_ptr(), dmods.data_ptr(), dtok.data_ptr()
        args.dhp = dhp.c()
        nxt = PlanesC()
        args.dhp_out = C_.pointer(nxt)
        reducing = self._exchange.reducing()
        in_c = self._armed is not None and self._side is not None and not self._exchange.active()
        # lab (PFPP_TRAIN_ADA_IN_C=1, single rank): the blocks' AdaLN-linear gradients (and, armed, their AdamW) per block on the side
        # stream instead of in the iteration's tail.  Measured SLOWER (6.10 -> 6.27 ms, profiles/r05f_ab_tail_ada_tables.txt): what ends the
        # iteration is the side stream's work for layer 0 (its grouped weight gradients + AdamW, ~260 us behind the chain), which the
        # tail's own launches used to cover — moving them onto that stream lengthens exactly the critical part.  Default off.
        ada_c = (self._side is not None and not self._exchange.active() and os.environ.get("PFPP_TRAIN_ADA_IN_C", "0") == "1")
        if ada_c:
            se = s["se"]
            dse = torch.empty_like(se)
            opsd = self.flat.operands()
            args.ada_se, args.ada_dse = se.data_ptr(), dse.data_ptr()
            args.ada_w, args.ada_gw, args.ada_gb = opsd["w"]["ada.w"].f32.data_ptr(), g["ada.w"].data_ptr(), g["ada.b"].data_ptr()
            args.ada_adamw_w, args.ada_adamw_b = (self._ada_static if in_c else (None, None))
            for t_ in (se, dse, dmods):
                t_.record_stream(self._side)
            s["_ada_dse"] = dse
        else:
            args.ada_se = args.ada_dse = args.ada_w = args.ada_gw = args.ada_gb = None
            args.ada_adamw_w = args.ada_adamw_b = None
        if in_c and os.environ.get("PFPP_TRAIN_TABLES_EARLY", "0") == "1" and not self._accumulated:
            # (not after a no_sync backward: rows an earlier micro-batch touched hold a gradient this launch would apply unscaled and zero)
            # lab (PFPP_TRAIN_TABLES_EARLY=1): the 12 timestep tables (a third of all parameters) — only the batch's rows get a gradient
            # this step, so every other row's AdamW update needs nothing of this backward and can go out now, on the side stream
            # (pfpp_adamw_rows; bit-identical to the one-pass update, tested).  Measured neutral (6.14 / 6.12 ms,
            # profiles/r05g_ab_tail_tables_dwsplit.txt): the tail is bounded by the side stream's layer-0 work, not by this launch.
            hp = self._armed
            f = self.flat
            n_tab = f.offset["transformer_layers.0.norm1.linear.weight"]
            tab_names = [n_ for n_ in f.order if n_.endswith(".emb.weight")]
            if not tab_names or min(f.offset[n_] for n_ in tab_names) != 0 or max(f.offset[n_] + f.view(f.params, n_).numel() for n_ in tab_names) != n_tab:
                raise RuntimeError("PFPP_TRAIN_TABLES_EARLY: the timestep tables are not the flat buffer's leading range [0, norm1.linear.weight)")
            shp = opsd["w"]["ada.tables"].shape if ada_c else self.flat.operands()["w"]["ada.tables"].shape
            t64 = s["t64"]
            views = [f_[:n_tab].view(shp) for f_ in (f.params, f.grads, f.exp_avg, f.exp_avg_sq)]
            self._run_on(self._side, lambda: T.adamw_rows(*views, t64, mode=0, lr=hp["lr"], beta1=hp["betas"][0], beta2=hp["betas"][1],
                                                          eps=hp["eps"], weight_decay=hp["weight_decay"], step=self.step_count + 1,
                                                          hi=f.hi[:n_tab], lo=f.lo[:n_tab], g_scale=1.0, zero_grad=self._armed_zero,
                                                          overflow=self._overflow))
            t64.record_stream(self._side)
            self._tables_rows = (t64, shp, n_tab)
        if in_c:
            # optimizer in the backward (arm_optimizer), single rank: each layer's AdamW is queued on the side stream by the C sequencer
            hp = self._armed
            step = self.step_count + 1
            args.adamw = adam
            args.lr, args.beta1, args.beta2, args.eps, args.weight_decay = hp["lr"], hp["betas"][0], hp["betas"][1], hp["eps"], hp["weight_decay"]
            args.bc1, args.bc2 = 1.0 - hp["betas"][0] ** step, 1.0 - hp["betas"][1] ** step
            args.opt_g_scale, args.opt_zero_grad = 1.0, int(self._armed_zero)
            args.overflow = self._overflow.data_ptr() if self._overflow is not None else None
        else:
            args.adamw = None
        side_arg = C_.c_void_p(side_h) if side_h is not None else None
        if not reducing:
            _lib.check(lib.pfpp_tlayers_bwd(C_.byref(args), 0, self.num_layers, C_.c_void_p(main_h), side_arg), "pfpp_tlayers_bwd")
            if in_c:
                self._early.extend(self.flat.layer_ranges)
                if ada_c:
                    self._early.extend(self._ada_ranges)
                if self._tables_rows is not None:
                    self._early.append((0, self._tables_rows[2]))
        else:
            for i in reversed(range(self.num_layers)):
                _lib.check(lib.pfpp_tlayers_bwd(C_.byref(args), i, i + 1, C_.c_void_p(main_h), side_arg), "pfpp_tlayers_bwd")
                args.dhp = nxt                         # the planes of the gradient entering layer i - 1 (inside the temporaries)
                self._layer_done(i)
        return dtok

    def _zeroed_dx_pool(self, M, C, w):
        """the split input-gradient GEMMs of the backward (3 per layer at token counts below ~8,000) add into zeroed outputs:
        one fill for all of them, issued on the weight-gradient stream while it has nothing else to do (the forward), instead of
        18 fills on the dependency chain of the backward -> (buffers [n, M, C], event) or None"""
        inner2 = w["0.ff1.w"].f32.shape[0]
        n = self.num_layers * ((2 if T.dx_splits(M, C, 3 * C) else 0) + (1 if T.dx_splits(M, C, inner2) else 0))
        if n == 0 or self._side is None or os.environ.get("PFPP_TRAIN_DX_POOL", "1") == "0":
            return None
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self._side):
            pool = torch.zeros((n, M, C), dtype=torch.float32, device=self.flat.params.device)
            ev = torch.cuda.Event()
            ev.record(self._side)
        pool.record_stream(main)
        return [pool, ev, 0]

    @staticmethod
    def _take_zeroed(pool, M, K_in, N_out):
        if pool is None or not T.dx_splits(M, K_in, N_out) or pool[2] >= pool[0].shape[0]:
            return None
        if pool[1] is not None:
            torch.cuda.current_stream().wait_event(pool[1])
            pool[1] = None
        pool[2] += 1
        return pool[0][pool[2] - 1]

    @staticmethod
    def _proj_residual(att, wo, bo, h, p, seed, site):
        M, C = h.shape
        if p > 0.0:
            y = ops.gemm(att, wo, M=M, N=C, K=C, lda=C, ldc=C, bias=bo)
            return T.dropout(y, p, seed, site, res=h, out=y)
        return ops.gemm(att, wo, M=M, N=C, K=C, lda=C, ldc=C, bias=bo, residual=h, ldr=C)

    # ------------------------------------------------------------------------------------------ backward
    def no_sync(self):
        """context manager for gradient accumulation across ranks: backwards inside it only accumulate locally; the first backward
        outside it all-reduces the accumulated sum (same contract as DistributedDataParallel.no_sync)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = prev
        return cm()

    def backward(self, ctx: TrainContext, dpred: torch.Tensor, amax: Optional[torch.Tensor] = None) -> None:
        """accumulate d(loss)/d(parameter) into the flat gradient buffer (= every parameter's .grad).  amax: max |dpred| on the device
        when the caller has it (the fused loss kernel), for the gradient-scale tracking"""
        if self._exchange.active() and self._exchanged:
            raise RuntimeError("DenoiserTrainEngine.backward: the flat gradient buffer was already all-reduced in place by a previous "
                               "backward of this step; a second backward would reduce the summed micro-batch again.  Accumulate with "
                               "`with engine.no_sync():` around every backward but the last, or call optimizer_step() in between")
        if _DIAG_HOST_DELAY_US:                 # diagnostic: is the iteration host-bound? (busy-wait on the host before the backward is enqueued)
            import time
            t_end = time.perf_counter() + _DIAG_HOST_DELAY_US * 1e-6
            while time.perf_counter() < t_end:
                pass
        self._join_side()
        self._heads_ev_fresh = False
        self._exchange.enabled = self._sync
        self._exchanged = self._exchange.active() and self._sync
        if self._exchange.active() and not self._sync:
            self._accumulated = True             # this step's table gradients are no longer "the rows of one batch": exchange them densely
        self.flat.attach_grads()
        self.flat._clean = False                 # gradients are about to be accumulated
        ops_ = self.flat.operands()
        w, g = ops_["w"], ops_["g"]
        s = ctx.t
        if self._dyn_gscale:
            self._update_grad_scale(dpred, amax)
        G = self.grad_scale
        B, L, C, Fv, M = s["B"], s["L"], s["C"], s["Fv"], s["M"]
        H = self.num_heads
        dh = C // H
        dev = dpred.device
        seed, p_lay, p_tok, fuse = s["seed"], s["p_lay"], s["p_tok"], s["fuse"]
        fused_heads = "heads_saved" in s
        dpred = _f32c(dpred).reshape(s["n_slots"], 7)
        dout_c = None if fused_heads else dpred[s["slot"]].contiguous()         # [Fv, 7] (the fused heads kernel gathers by slot itself)

        # every small zero-initialised buffer of the backward out of ONE zeroed arena (one fill launch instead of eight)
        ld_sf, ld_pf = (s["sf"].shape[1], s["pf"].shape[1]) if s["sf"] is not None else (0, 0)
        h1 = C // 2                                     # hidden width of the heads' second linear
        sizes = [Fv * C, Fv * 4, Fv * 4, 4 * h1, 4 * h1, s["mods"].numel(), C * ld_sf, C * ld_pf]
        offs = [0]
        for n_ in sizes:
            offs.append(offs[-1] + (n_ + 3) // 4 * 4)
        arena = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        carve = lambda k, *shape: arena[offs[k]: offs[k] + sizes[k]].view(*shape)
        dpads, dw4s = [carve(1, Fv, 4), carve(2, Fv, 4)], [carve(3, 4, h1), carve(4, 4, h1)]

        # ---- output heads (denoiser_transformer.py:138-147)
        if fused_heads:
            dh_ = self._heads_backward_fused(s, w, g, dpred, G, Fv, L)
        else:
            dh_ = self._heads_backward_layerwise(s, w, g, dout_c, G, Fv, L, C, carve, dpads, dw4s)

        dmods = carve(5, *s["mods"].shape)
        # multi-rank: the two AdaLN linears of a block get their gradients as soon as the block's backward is through and travel with
        # the block's slice (otherwise 25 MB of dense gradient would be left for the exposed tail after the backward)
        self._ada_layerwise = (dmods, s["se"], g, B, C, G) if (self._exchange.reducing() and self._ada_per_layer) else None
        if self._planes and s.get("cseq") is not None:
            dtok = self._backward_layers_c(s, w, g, dh_, dmods)
        elif self._planes:
            dtok = self._backward_layers_planes(s, w, g, dh_, dmods)
        else:
            dtok = self._backward_layers(s, w, g, dh_, dmods)

        # ---- tokens (denoiser_transformer.py:117-135,150-156,173-185)
        if p_tok > 0.0 and not fuse:
            dtok = T.dropout(dh_, p_tok, seed, 0)
        if s.get("ft") is not None:
            # all five gradients of the embedding from one contraction over the tokens (csrc/embed_train.hip)
            T.token_embed_bwd(dtok, *s["ft"], g["shape.w"], g["shape.b"], g["param.w"], g["param.b"], g["ref_emb"], Fv, L, g_scale=G)
        else:
            dws = carve(6, C, ld_sf)
            T.grad_weight(dtok, s["sf"], dws, g_scale=G)
            g["shape.w"].add_(dws[:, : g["shape.w"].shape[1]])
            T.colsum(dtok, g["shape.b"])
            dx_emb = T.token_combine_bwd(dtok, s["ref_u8"], g["ref_emb"], L, slot=s["slot32"])
            dwp = carve(7, C, ld_pf)
            T.grad_weight(dx_emb, s["pf"], dwp, g_scale=G)
            g["param.w"].add_(dwp[:, : g["param.w"].shape[1]])
            T.colsum(dx_emb, g["param.b"])

        # ---- AdaLN modulation (attention.py:21-25): mods[j] = silu(table_j[t]) . W_j^T + b_j
        n_ada = 2 * self.num_layers
        se = s["se"]
        dse = s.pop("_ada_dse", None)
        if dse is not None:
            # the C sequencer queued the AdaLN linears' gradients and d/d(embedded timestep) per block on the side stream
            torch.cuda.current_stream().wait_stream(self._side)
        elif self._ada_layerwise is None and self._ada_bwd_fused and C % 32 == 0:
            # both gradients and d/d(embedded timestep) of the twelve linears in two fp32 launches (csrc/ada_bwd.hip)
            dse = T.ada_linear_bwd(dmods, se, w["ada.w"].f32.view(n_ada, 2 * C, C), g["ada.w"].view(n_ada, 2 * C, C), g["ada.b"])
        else:
            if self._ada_layerwise is None:
                T.colsum(dmods, g["ada.b"], rows=B, cols=2 * C, ld=2 * C, batch=n_ada, sx=B * 2 * C, so=2 * C)
                T.gemm_grad(dmods, se, g["ada.w"], M=2 * C, N=C, K=B, lda=2 * C, ldw=C, ldc=C, a_kmajor=True, w_kmajor=True,
                            accumulate=True, batch=n_ada, sA=B * 2 * C, sW=B * C, sC=2 * C * C, a_scale=G)
            dse = torch.empty_like(se)
            T.gemm_grad(dmods, w["ada.w"].f32, dse, M=B, N=C, K=2 * C, lda=2 * C, ldw=C, ldc=C, w_kmajor=True, batch=n_ada,
                        sA=B * 2 * C, sW=2 * C * C, sC=B * C, a_scale=G)
        self._ada_layerwise = None
        if self._sparse_tables and self._exchange.reducing() and not self._accumulated:
            dse_all, t_all = self._exchange.gather_rows(dse, s["t64"], dim=1)       # [n_ada, world*B, C], [world*B]
            T.silu_embed_bwd(w["ada.tables"], t_all, dse_all, g["ada.tables"], active=self._tab_active)
        else:
            if self._exchange.active() and self._tab_active is not None:
                # the dense all-reduce brings in rows other ranks indexed, which nobody marks here: every row counts from now on
                self._tab_active.fill_(-1)
            T.silu_embed_bwd(w["ada.tables"], s["t64"], dse, g["ada.tables"], active=self._tab_active)
        self._all_done()
        # the main stream now waits for every reader of this step's arenas: hand them back for the next step (stream order protects them)
        if "_bwd_tmp" in s:
            self._give_arena(*s.pop("_bwd_tmp"))
        ctx.release()

    def _head_structs(self, w, g):
        """(trans, rot, g_trans, g_rot) ctypes structs over the flat buffers (built once: the buffers never move)"""
        if self._heads_static is None:
            from ._lib import HeadGrads

            hp, hg = [], []
            for n in ("mlp_out_trans", "mlp_out_rot"):
                hp.append(T.head_params(w[f"{n}.0.w"], w[f"{n}.2.w"], w[f"{n}.4.w"].f32, w[f"{n}.0.b"], w[f"{n}.2.b"], w[f"{n}.4.b"]))
                hg.append(HeadGrads(g[f"{n}.4.w"].data_ptr(), g[f"{n}.4.b"].data_ptr(), g[f"{n}.2.b"].data_ptr(), g[f"{n}.0.b"].data_ptr()))
            self._heads_static = (hp[0], hp[1], hg[0], hg[1])
        return self._heads_static

    def _heads_backward_fused(self, s, w, g, dout_c, G, Fv, L):
        """one launch for the chain (dW4 / every bias gradient / da1 / da0 / d pooled of both heads, + the mean-pool backward), one
        grouped launch for the four wide weight gradients (off the critical chain: on the weight-gradient stream) -> d/dh [M, C]"""
        trans, rot, g_trans, g_rot = self._head_structs(w, g)
        a0, v0, a1, v1 = s["heads_saved"]
        da0, da1, dh_ = T.heads_bwd(dout_c, trans, rot, s["heads_saved"], g_trans, g_rot, G, L, slot=s["slot32"])
        pooled = s["pooled"]
        problems = [(da1[0], v0[0], g["mlp_out_trans.2.w"]), (da1[1], v0[1], g["mlp_out_rot.2.w"]),
                    (da0[0], pooled, g["mlp_out_trans.0.w"]), (da0[1], pooled, g["mlp_out_rot.0.w"])]
        if self._side is None:
            T.grad_weight_group(problems, g_scale=G)
        else:
            self._run_on(self._side, lambda: T.grad_weight_group(problems, g_scale=G))
            for t_ in (da0, da1, v0, pooled):
                t_.record_stream(self._side)
            if self._heads_dw_ev is None:
                self._heads_dw_ev = torch.cuda.Event()
            self._heads_dw_ev.record(self._side)          # (the closing AdamW waits for this instead of for the whole stream: _all_done)
            self._heads_ev_fresh = True
        return dh_

    def _heads_backward_layerwise(self, s, w, g, dout_c, G, Fv, L, C, carve, dpads, dw4s):
        dpooled = carve(0, Fv, C)
        for hi_, (name, c0, width) in enumerate((("mlp_out_trans", 0, 3), ("mlp_out_rot", 3, 4))):
            a0, v0, a1, v1 = s["heads"][name]
            dpad = dpads[hi_]
            dpad[:, :width] = dout_c[:, c0:c0 + width]
            T.colsum(dout_c, g[f"{name}.4.b"], rows=Fv, cols=width, ld=7, x_off=c0)
            dw4 = dw4s[hi_]
            T.grad_weight(dpad, v1, dw4, g_scale=G)
            g[f"{name}.4.w"].add_(dw4[:width])
            dv1 = T.gemm_grad(dpad, w[f"{name}.4.w"].f32, torch.empty_like(v1), M=Fv, N=v1.shape[1], K=width, lda=4,
                              ldw=v1.shape[1], ldc=v1.shape[1], w_kmajor=True, a_scale=G)
            da1 = T.act_bwd(a1, dv1, "silu")
            self._linear_bwd(da1, v0, w[f"{name}.2.w"], g[f"{name}.2.w"], g[f"{name}.2.b"])
            dv0 = T.grad_input(da1, w[f"{name}.2.w"].f32, g_scale=G)
            da0 = T.act_bwd(a0, dv0, "silu")
            self._linear_bwd(da0, s["pooled"], w[f"{name}.0.w"], g[f"{name}.0.w"], g[f"{name}.0.b"])
            T.gemm_grad(da0, w[f"{name}.0.w"].f32, dpooled, M=Fv, N=C, K=da0.shape[1], lda=da0.shape[1], ldw=C, ldc=C,
                        w_kmajor=True, accumulate=True, split_k=1, a_scale=G)
        return T.mean_pool_bwd(dpooled, L)                                                 # running d/dh [M, C]

    def _backward_layers(self, s, w, g, dh_, dmods):
        """transformer blocks, register-staged backward GEMMs (csrc/gemm_grad.hip) -> d/d(tokens)"""
        G = self.grad_scale
        L, C, Fv = s["L"], s["C"], s["Fv"]
        H = self.num_heads
        dh = C // H
        seed, p_lay, p_tok, fuse = s["seed"], s["p_lay"], s["p_tok"], s["fuse"]
        pool = s.get("dx_pool")
        for i in reversed(range(self.num_layers)):
            lay = s["layers"][i]
            # ---- feed-forward (attention.py:87-90)
            self._linear_bwd(dh_, lay["u"], w[f"{i}.ff2.w"], g[f"{i}.ff2.w"], g[f"{i}.ff2.b"], guard=True)
            du = T.grad_input(dh_, w[f"{i}.ff2.w"].f32, g_scale=G)
            dz = T.geglu_bwd(lay["z"], du, p_lay, seed, 3 + 3 * i)
            del du
            self._linear_bwd(dz, lay["n3"], w[f"{i}.ff1.w"], g[f"{i}.ff1.w"], g[f"{i}.ff1.b"])
            dn = T.grad_input(dz, w[f"{i}.ff1.w"].f32, g_scale=G, zeroed=self._take_zeroed(pool, dz.shape[0], C, dz.shape[1]))
            del dz
            self._before_inplace_update()
            dy = T.layernorm_bwd(lay["h2"], dn, dh_, gamma=w[f"{i}.norm3.g"], group_rows=32, dmult=g[f"{i}.norm3.g"],
                                 dadd=g[f"{i}.norm3.b"], ld_d=0, drop=(p_lay, seed, 2 + 3 * i) if fuse and p_lay > 0.0 else None)
            # ---- global attention (attention.py:82-85)
            if p_lay > 0.0 and not fuse:
                dy = T.dropout(dh_, p_lay, seed, 2 + 3 * i)
            self._linear_bwd(dy, lay["att2"], w[f"{i}.global_attn.o.w"], g[f"{i}.global_attn.o.w"], g[f"{i}.global_attn.o.b"],
                             guard=dy is dh_)
            datt = T.grad_input(dy, w[f"{i}.global_attn.o.w"].f32, g_scale=G)
            dqkv = T.attn_dense_bwd(lay["qkv2"], lay["att2"], datt, lay["lse"], s["seq_off"], s["seq_len"], s["max_len"], H, dh,
                                    s["att_scale"])
            self._linear_bwd(dqkv, lay["n2"], None, g[f"{i}.global_attn.qkv.w"], None)
            dn = T.grad_input(dqkv, w[f"{i}.global_attn.qkv.w"].f32, g_scale=G, zeroed=self._take_zeroed(pool, dqkv.shape[0], C, 3 * C))
            self._before_inplace_update()
            dy = T.layernorm_bwd(lay["h1"], dn, dh_, mod=s["mods"][2 * i + 1], group_batch=s["frag_b"], group_rows=L,
                                 dmult=dmods[2 * i + 1], dadd=dmods[2 * i + 1][:, C:], ld_d=2 * C,
                                 drop=(p_lay, seed, 1 + 3 * i) if fuse and p_lay > 0.0 else None)
            # ---- self attention (attention.py:77-80)
            if p_lay > 0.0 and not fuse:
                dy = T.dropout(dh_, p_lay, seed, 1 + 3 * i)
            self._linear_bwd(dy, lay["att1"], w[f"{i}.self_attn.o.w"], g[f"{i}.self_attn.o.w"], g[f"{i}.self_attn.o.b"],
                             guard=dy is dh_)
            datt = T.grad_input(dy, w[f"{i}.self_attn.o.w"].f32, g_scale=G)
            dqkv = T.attn_blockdiag_bwd(lay["qkv1"], datt, Fv, L, H, dh, s["att_scale"])
            self._linear_bwd(dqkv, lay["n1"], None, g[f"{i}.self_attn.qkv.w"], None)
            dn = T.grad_input(dqkv, w[f"{i}.self_attn.qkv.w"].f32, g_scale=G, zeroed=self._take_zeroed(pool, dqkv.shape[0], C, 3 * C))
            self._before_inplace_update()
            dtok = T.layernorm_bwd(lay["h0"], dn, dh_, mod=s["mods"][2 * i], group_batch=s["frag_b"], group_rows=L,
                                   dmult=dmods[2 * i], dadd=dmods[2 * i][:, C:], ld_d=2 * C,
                                   drop=(p_tok, seed, 0) if i == 0 and fuse and p_tok > 0.0 else None)
            self._layer_done(i)

        return dtok

    def _dw_planes(self, dyp, xp, gw, gb) -> None:
        """dW += dY^T . X (both operands read in place as k-major planes), db += colsum(dY) — on the side stream"""
        from . import planes as P

        if self._dw_pending is not None:       # bench.py's traced pass: the block's weight gradients go out as ONE launch, like the C sequencer's
            self._dw_pending.append((dyp, xp, gw, gb if (gb is not None and gb.is_contiguous()) else None))
            if gb is not None and not gb.is_contiguous():
                P.colsum(dyp, gb)
            return

        def issue():
            fused = gb is not None and self._fuse_colsum and gb.is_contiguous()      # the bias gradient rides in the dW kernel
            P.gemm(dyp, xp, gw, M=gw.shape[0], N=gw.shape[1], K=dyp.shape[0], a_kmajor=True, w_kmajor=True, accumulate=True,
                   colsum=gb if fused else None)
            if gb is not None and not fused:
                P.colsum(dyp, gb)

        if self._side is None:
            issue()
            return
        st = self._side
        self._run_on(st, issue)
        dyp.record_stream(st)
        xp.record_stream(st)

    def _dw_flush(self, K: int) -> None:
        """the collected weight gradients of a block as one pfpp_gemm_dw_group launch (side stream when there is one)"""
        from . import planes as P

        jobs, self._dw_pending = self._dw_pending, []
        if not jobs:
            return
        if self._side is None:
            P.dw_group(jobs, K)
            return
        self._run_on(self._side, lambda: P.dw_group(jobs, K))
        for dyp, xp, _, _ in jobs:
            dyp.record_stream(self._side)
            xp.record_stream(self._side)

    def _run_on(self, st, fn) -> None:
        """fn() only launches pfpp kernels into existing buffers: send them to stream `st`, ordered after everything queued on the
        current stream so far — two direct HIP calls and the wrappers' stream override instead of torch's wait_stream + stream
        context (pfpp_hip.hipstream: ~30 us of Python per use, ~50 uses per backward)"""
        from . import hipstream as HS

        h = st.cuda_stream
        try:
            HS.wait_for(h, ops.raw_stream_id(self.flat.params.device.index))
        except (RuntimeError, OSError):
            # no HIP runtime handle to call directly (a torch build on a system ROCm we cannot dlopen): torch's own ordering
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fn()
            return
        prev, ops.STREAM_OVERRIDE = ops.STREAM_OVERRIDE, h
        try:
            fn()
        finally:
            ops.STREAM_OVERRIDE = prev

    def _backward_layers_planes(self, s, w, g, dh_, dmods):
        """transformer blocks on the plane GEMM: dX = dY . W reads the weight planes in place as the k-major operand, dW = dY^T . X
        reads both dY and the saved activation planes in place; the kernels between them (GEGLU / LayerNorm / attention backward)
        hand dY over as planes of grad_scale * dY -> d/d(tokens)"""
        from . import planes as P

        G = self.grad_scale
        L, C, Fv, M = s["L"], s["C"], s["Fv"], s["M"]
        H = self.num_heads
        dh = C // H
        dev = dh_.device
        seed, p_lay, p_tok, fuse = s["seed"], s["p_lay"], s["p_tok"], s["fuse"]
        frag_b = s["frag_b"]

        def wp(key):
            pw = w[key]
            return P.Planes(pw.hi, pw.lo, pw.scale)

        frags = self._traced_frags(w, transposed=True) if _traced_wd() else {}

        def dx(dyp, key, n_in):
            out = torch.empty((M, n_in), dtype=torch.float32, device=dev)
            if _traced_wd() and not key.endswith("ff1.w") and n_in % 128 == 0 and dyp.shape[1] % 64 == 0:
                return P.gemm_wd(dyp, wp(key), out, M=M, N=n_in, K=dyp.shape[1], transposed=True, frag=frags.get(key))
            return P.gemm(dyp, wp(key), out, M=M, N=n_in, K=dyp.shape[1], w_kmajor=True)

        drop_lay = fuse and p_lay > 0.0
        dhp = P.split(dh_, G)                         # d/dh of the last block's output: dY of its second feed-forward linear
        dtok = None
        # the traced pass of bench.py takes the launches the C sequencer takes: weight-direct GEMMs (dx above) and the grouped
        # weight-gradient launch; untraced, this sequence stays the per-weight cross-check
        self._dw_pending = [] if (ops.GEMM_TRACE is not None and os.environ.get("PFPP_TRAIN_DW_GROUP", "1") != "0") else None
        for i in reversed(range(self.num_layers)):
            lay = s["layers"][i]
            inner = lay["u"].shape[1]
            # ---- feed-forward (attention.py:87-90)
            self._dw_planes(dhp, lay["u"], g[f"{i}.ff2.w"], g[f"{i}.ff2.b"])
            du = dx(dhp, f"{i}.ff2.w", inner)
            dzp = T.geglu_bwd_planes(lay["z"], du, p_lay, seed, 3 + 3 * i, G)
            del du
            self._dw_planes(dzp, lay["n3"], g[f"{i}.ff1.w"], g[f"{i}.ff1.b"])
            dn = dx(dzp, f"{i}.ff1.w", C)
            del dzp
            dyp, _, _ = T.layernorm_bwd_planes(lay["h2"], dn, dh_, G, gamma=w[f"{i}.norm3.g"], group_rows=32, dmult=g[f"{i}.norm3.g"],
                                               dadd=g[f"{i}.norm3.b"], ld_d=0, drop=(p_lay, seed, 2 + 3 * i) if drop_lay else None)
            if p_lay > 0.0 and not fuse:
                dyp = P.split(T.dropout(dh_, p_lay, seed, 2 + 3 * i), G)
            # ---- global attention (attention.py:82-85)
            self._dw_planes(dyp, lay["att2p"], g[f"{i}.global_attn.o.w"], g[f"{i}.global_attn.o.b"])
            datt = dx(dyp, f"{i}.global_attn.o.w", C)
            dqkvp = T.attn_dense_bwd_planes(lay["qkv2"], lay["att2"], datt, lay["lse"], s["seq_off"], s["seq_len"], s["max_len"], H, dh,
                                            s["att_scale"], G)
            self._dw_planes(dqkvp, lay["n2"], g[f"{i}.global_attn.qkv.w"], None)
            dn = dx(dqkvp, f"{i}.global_attn.qkv.w", C)
            dyp, _, _ = T.layernorm_bwd_planes(lay["h1"], dn, dh_, G, mod=s["mods"][2 * i + 1], group_batch=frag_b, group_rows=L,
                                               dmult=dmods[2 * i + 1], dadd=dmods[2 * i + 1][:, C:], ld_d=2 * C,
                                               drop=(p_lay, seed, 1 + 3 * i) if drop_lay else None)
            if p_lay > 0.0 and not fuse:
                dyp = P.split(T.dropout(dh_, p_lay, seed, 1 + 3 * i), G)
            # ---- self attention (attention.py:77-80)
            self._dw_planes(dyp, lay["att1"], g[f"{i}.self_attn.o.w"], g[f"{i}.self_attn.o.b"])
            datt = dx(dyp, f"{i}.self_attn.o.w", C)
            dqkvp = T.attn_blockdiag_bwd_planes(lay["qkv1"], datt, Fv, L, H, dh, s["att_scale"], G)
            self._dw_planes(dqkvp, lay["n1"], g[f"{i}.self_attn.qkv.w"], None)
            dn = dx(dqkvp, f"{i}.self_attn.qkv.w", C)
            if i > 0:
                # the updated running gradient is the dY of block i-1's second feed-forward linear
                _, dhp, _ = T.layernorm_bwd_planes(lay["h0"], dn, dh_, G, mod=s["mods"][2 * i], group_batch=frag_b, group_rows=L,
                                                   dmult=dmods[2 * i], dadd=dmods[2 * i][:, C:], ld_d=2 * C, want_ret=False, want_dx=True)
            else:
                dtok = T.layernorm_bwd(lay["h0"], dn, dh_, mod=s["mods"][0], group_batch=frag_b, group_rows=L, dmult=dmods[0],
                                       dadd=dmods[0][:, C:], ld_d=2 * C, drop=(p_tok, seed, 0) if fuse and p_tok > 0.0 else None)
            if self._dw_pending is not None:
                self._dw_flush(M)
            self._layer_done(i)
        self._dw_pending = None
        return dtok

    def _linear_bwd(self, dy, x, wpw, gw, gb, guard: bool = False) -> None:
        """dW += dy^T x, db += colsum(dy) — on the side stream when there is one.  `guard`: the caller goes on to update
        dy in place on the main stream."""
        fused_db = gb if (self._fuse_colsum and gb is not None and gb.is_contiguous()) else None    # bias sums ride in the dW GEMM
        if self._side is None:
            T.grad_weight(dy, x, gw, g_scale=self.grad_scale, db=fused_db)
            if gb is not None and fused_db is None:
                T.colsum(dy, gb)
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)                 # dy (and x) are ready once everything queued so far has run
        with torch.cuda.stream(self._side):
            T.grad_weight(dy, x, gw, g_scale=self.grad_scale, db=fused_db)
            if gb is not None and fused_db is None:
                T.colsum(dy, gb)
        dy.record_stream(self._side)                 # the caching allocator must not recycle them under the side stream
        x.record_stream(self._side)
        if guard:                                    # the caller goes on to UPDATE dy in place on the main stream
            self._dy_read = torch.cuda.Event()
            self._dy_read.record(self._side)

    def _before_inplace_update(self) -> None:
        """the running residual-stream gradient is read by weight-gradient GEMMs on the side stream and then updated in
        place (dx += LayerNorm backward) on the main stream: the update waits for those reads (write-after-read across
        streams — record_stream only covers the allocator, not an explicit in-place write)"""
        ev = getattr(self, "_dy_read", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._dy_read = None

    # ------------------------------------------------------------------------------------------ data parallel
    def _layer_done(self, i: int) -> None:
        """gradients of layer i are final: start their all-reduce while the earlier layers still compute, and — when the optimizer is
        armed (arm_optimizer) — give the layer its AdamW update as soon as its gradients are final (N = 1) / reduced (N > 1)."""
        reducing = self._exchange.reducing()
        if self._armed is not None and self._side is not None and not self._exchange.active():
            # optimizer in the backward (arm_optimizer): this layer's slice of the flat buffer is final once its weight
            # gradients (side stream) and LayerNorm gradients (main stream, all queued by now) have run — update it on the side
            # stream under the remaining backward instead of in the 0.3 ms AdamW launch that runs alone at the iteration's end
            armed = self._armed
            self._run_on(self._side, lambda: self._adamw_range(*self.flat.layer_ranges[i], step=self.step_count + 1, g_scale=1.0,
                                                               zero_grad=self._armed_zero, **armed))
            self._early.append(self.flat.layer_ranges[i])
        if reducing:
            extra = ()
            if self._ada_layerwise is not None:
                dmods, se, g, B, C, G = self._ada_layerwise
                j = 2 * i
                T.colsum(dmods[j:j + 2], g["ada.b"][j:j + 2], rows=B, cols=2 * C, ld=2 * C, batch=2, sx=B * 2 * C, so=2 * C)
                T.gemm_grad(dmods[j:j + 2], se[j:j + 2], g["ada.w"][j:j + 2], M=2 * C, N=C, K=B, lda=2 * C, ldw=C, ldc=C, a_kmajor=True,
                            w_kmajor=True, accumulate=True, batch=2, sA=B * 2 * C, sW=B * C, sC=2 * C * C, a_scale=G)
                f = self.flat
                t0 = "transformer_layers.0"
                aw, ab = f.offset[f"{t0}.norm1.linear.weight"], f.offset[f"{t0}.norm1.linear.bias"]
                extra = ((aw + j * 2 * C * C, aw + (j + 2) * 2 * C * C), (ab + j * 2 * C, ab + (j + 2) * 2 * C))
            # the layer's slice holds gradients from BOTH streams (weights / biases: side stream; LayerNorm gamma / beta and the
            # AdaLN linears: main stream), so the collective is ordered after everything queued on either of them so far.  It is
            # issued from a third stream: the next layer's weight-gradient GEMMs (side stream) must not queue up behind it.
            comm = self._comm_stream()
            comm.wait_stream(torch.cuda.current_stream())
            if self._side is not None:
                comm.wait_stream(self._side)
            with torch.cuda.stream(comm):
                self._exchange.layer_done(i, extra)
                if self._armed is not None:
                    # N > 1 keeps the benchmarked schedule: the layer's AdamW runs BEHIND its all-reduce on the
                    # exchange stream, under the rest of the backward; only embeddings / tables / heads are left for optimizer_step
                    self._exchange.wait_pending()                # RCCL: the exchange stream waits, the host does not
                    # (the AdaLN linears that travelled with the layer are NOT updated here: the end of the backward still reads
                    # their weights for d/d(timestep embedding))
                    a, b = self.flat.layer_ranges[i]
                    self._adamw_range(a, b, step=self.step_count + 1, g_scale=self._exchange.mean_factor(),
                                      zero_grad=self._armed_zero, **self._armed)
                    self._early.append((a, b))

    def _comm_stream(self):
        """the stream the per-layer gradient all-reduces (and the AdamW launches behind them) are issued from (N > 1 only)"""
        st = getattr(self, "_comm", None)
        if st is None:
            st = self._comm = torch.cuda.Stream(device=self.flat.params.device)
        return st

    def _join_side(self) -> None:
        """the caller's stream waits for the weight-gradient stream if the last backward left that to the optimizer step"""
        if self._join_pending:
            self._join_pending = False
            torch.cuda.current_stream().wait_stream(self._side)

    def _all_done(self) -> None:
        # armed single-rank step through the C sequencer: what is still running on the weight-gradient stream when the chain ends is the
        # first block's weight gradients and update (~0.1 ms), and nothing the closing AdamW of optimizer_step() touches — tables,
        # AdaLN linears, embeddings, heads — comes from there (the heads' wide weight gradients were queued first on that stream:
        # an event).  The join therefore moves behind those launches (optimizer_step), which then run under the stream's tail instead
        # of behind it; forward() joins too, should a caller skip the step.  PFPP_TRAIN_TAIL_OVERLAP=0: join here, as before
        if (self._side is not None and self._early and self._tail_overlap and self._heads_ev_fresh and getattr(self, "_comm", None) is None
                and not self._exchange.active()):
            self._join_pending = True
            self._exchange.all_done(dense=self._accumulated)
            return
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if getattr(self, "_comm", None) is not None:
            torch.cuda.current_stream().wait_stream(self._comm)      # per-layer exchanges (and the updates behind them) join here
        self._exchange.all_done(dense=self._accumulated)

    def finish_grad_exchange(self) -> float:
        """wait for the gradient all-reduces; returns the factor that turns the summed gradients into the mean"""
        return self._exchange.finish()

    # ------------------------------------------------------------------------------------------ optimizer
    def arm_optimizer(self, *, lr: float = 2e-4, betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6,
                      zero_grad: bool = False) -> None:
        """optimizer-in-backward for the NEXT backward: every transformer layer's parameters take their AdamW update as soon as
        the layer's gradients are final (on the weight-gradient stream, under the rest of the backward); optimizer_step() with
        the same hyper-parameters then updates only what is left (embeddings, AdaLN tables and linears, output heads) and
        closes the step.  Same arithmetic per element as one launch over the flat buffer.  With N > 1 ranks the layer's update is
        queued behind the layer's all-reduce on the exchange stream (1 / world folded in), so the multi-rank step keeps the
        schedule the single-rank line is measured on.  One-shot; ignored (everything happens in optimizer_step) without a second
        stream at N = 1, and for backward passes under no_sync() (gradient accumulation: nothing is final yet)."""
        self._armed = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))
        self._armed_zero = bool(zero_grad)          # the per-layer updates also clear their gradients (optimizer_step(zero_grad=True))
        self._early = []
        self._tables_rows = None

    def _adamw_range(self, a: int, b: int, *, step: int, g_scale: float, lr, betas, eps, weight_decay, zero_grad: bool = False) -> None:
        f = self.flat
        T.adamw(f.params[a:b], f.grads[a:b], f.exp_avg[a:b], f.exp_avg_sq[a:b], lr=lr, beta1=betas[0], beta2=betas[1], eps=eps,
                weight_decay=weight_decay, step=step, hi=f.hi[a:b], lo=f.lo[a:b], g_scale=g_scale, zero_grad=zero_grad,
                overflow=self._overflow)

    def _adamw_tables_active(self, a: int, b: int, dense_ok: bool, g_scale: float, zero_grad: bool, hp) -> int:
        """the AdamW update of [a, b) is about to be issued: when the range starts with the timestep tables and their active-row bitmap is
        kept, update the tables through it (rows that never received a gradient are exactly unchanged: not touched) and return where
        the rest of the range starts; otherwise a (nothing done)"""
        if self._tab_active is None or a != 0 or not dense_ok:
            return a
        f = self.flat
        n_tab = f.offset["transformer_layers.0.norm1.linear.weight"]
        if b < n_tab:
            return a
        n_emb, C_ = f.named["transformer_layers.0.norm1.emb.weight"].shape
        shp = (n_tab // (n_emb * C_), n_emb, C_)
        T.adamw_rows_active(*(f_[:n_tab].view(shp) for f_ in (f.params, f.grads, f.exp_avg, f.exp_avg_sq)), self._tab_active, lr=hp["lr"],
                            beta1=hp["betas"][0], beta2=hp["betas"][1], eps=hp["eps"], weight_decay=hp["weight_decay"], step=self.step_count,
                            hi=f.hi[:n_tab], lo=f.lo[:n_tab], g_scale=g_scale, zero_grad=zero_grad, overflow=self._overflow)
        return n_tab

    def optimizer_step(self, *, lr: float = 2e-4, betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6,
                       zero_grad: bool = False) -> None:
        """AdamW over the flat buffer (configure_optimizers, denoiser.py:230-237) — one launch, or the ranges that an armed
        backward (arm_optimizer) has not updated yet.  zero_grad: also clear the gradients in that launch (the
        optimizer.step(); optimizer.zero_grad() pair of a training loop as one pass; the next flat.zero_grad() is then free)"""
        g_scale = self.finish_grad_exchange()
        self._exchanged = False
        self._accumulated = False
        self.step_count += 1
        f = self.flat
        hp = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))
        early, self._early = self._early, []
        armed, self._armed = self._armed, None
        rows, self._tables_rows = self._tables_rows, None
        if early:
            if armed != hp:
                raise RuntimeError("optimizer_step: hyper-parameters differ from the ones the backward was armed with")
            if zero_grad != self._armed_zero:
                raise RuntimeError("optimizer_step: zero_grad differs from what the backward was armed with")
            if rows is not None:
                # the timestep tables: every row but the batch's was updated at the start of the backward; now the batch's rows
                t64, shp, n_tab = rows
                T.adamw_rows(*(f_[:n_tab].view(shp) for f_ in (f.params, f.grads, f.exp_avg, f.exp_avg_sq)), t64, mode=1, lr=hp["lr"],
                             beta1=hp["betas"][0], beta2=hp["betas"][1], eps=hp["eps"], weight_decay=hp["weight_decay"],
                             step=self.step_count, hi=f.hi[:n_tab], lo=f.lo[:n_tab], g_scale=g_scale, zero_grad=zero_grad,
                             overflow=self._overflow)
            if self._join_pending and self._heads_dw_ev is not None:
                torch.cuda.current_stream().wait_event(self._heads_dw_ev)
            pos, total = 0, f.params.numel()
            for a, b in sorted(early) + [(total, total)]:
                if a > pos:
                    pos = self._adamw_tables_active(pos, a, rows is None, g_scale, zero_grad, hp)
                if a > pos:
                    self._adamw_range(pos, a, step=self.step_count, g_scale=g_scale, zero_grad=zero_grad, **hp)
                pos = max(pos, b)
            f._clean = bool(zero_grad)
        else:
            pos = self._adamw_tables_active(0, f.params.numel(), True, g_scale, zero_grad, hp)
            self._adamw_range(pos, f.params.numel(), step=self.step_count, g_scale=g_scale, zero_grad=zero_grad, **hp)
            f._clean = bool(zero_grad)
        self._join_side()                 # (deferred by _all_done: the launches above ran under the weight-gradient stream's tail)
        self._after_step_overflow()
        f.after_optimizer_step()
        cache = getattr(self.module, "_cache", None)
        if cache is not None:
            cache._key = None            # the eval-mode packing of the module is stale now

    def _after_step_overflow(self) -> None:
        """end of an optimizer step (main stream, every AdamW launch of the step queued): hand the step's overflow flag to the host
        through pinned memory and clear it for the next step; the flag of the step before last is read here (its copy has long
        finished: no stall) and drives the back-off of the gradient scale"""
        if self._overflow is None:
            return
        if self._ovf_ring is None:
            self._ovf_ring = [(torch.zeros(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
        host, ev = self._ovf_ring[self.step_count % 2]
        if self.step_count > 2:
            ev.synchronize()
            if int(host[0]) != 0:
                self.overflow_steps += 1
                self._backoff = max(2.0 ** -24, self._backoff / 16.0)
                self._clean_steps = 0
                self._apply_backoff()
            else:
                self._clean_steps += 1
                if self._clean_steps >= 200 and self._backoff < 1.0:
                    self._backoff, self._clean_steps = self._backoff * 2.0, 0
        host.copy_(self._overflow, non_blocking=True)
        ev.record()
        self._overflow.zero_()

    def _apply_backoff(self) -> None:
        if not self._dyn_gscale:                     # a pinned scale is lowered directly (the dynamic one is recomputed every backward)
            self.grad_scale = max(1.0, self.grad_scale / 16.0)

    def _update_grad_scale(self, dpred: torch.Tensor, amax: Optional[torch.Tensor] = None) -> None:
        if self._amax_ring is None:
            self._amax_ring = [(torch.zeros(1, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
        host, ev = self._amax_ring[self._n_backward % 2]
        if self._n_backward >= 2:
            ev.synchronize()                         # recorded two backward passes ago
            amax = float(host[0])
            if math.isfinite(amax) and amax > 0.0:
                self.grad_scale = max(1.0, float(2.0 ** min(40, max(0, 3 - math.floor(math.log2(amax))))) * self._backoff)
        # amax: max |dpred| already on the device (pfpp_mse_loss_masked wrote it next to dpred) — otherwise an abs + max over dpred
        host.copy_(amax if amax is not None else dpred.detach().abs().max().reshape(1), non_blocking=True)
        ev.record()
        self._n_backward += 1

    # ------------------------------------------------------------------------------------------ whole step
    def loss_and_grads(self, x, timesteps, latent, xyz, part_valids, scale, ref_part, noise, *, seed: int = 0,
                       train: bool = True, between=None) -> torch.Tensor:
        """forward + Denoiser._loss (denoiser.py:118-126) + backward; returns the loss [1].  between(): called once the forward is
        enqueued, before the backward (a scheduling hook: e.g. issue the next batch's encoder there)"""
        pred, ctx = self.forward(x, timesteps, latent, xyz, part_valids, scale, ref_part, seed=seed, train=train)
        if between is not None:
            between()
        n = pred.shape[0] * pred.shape[1]
        # the selection (valid & ~reference) is evaluated inside the loss kernel, which also leaves max |dpred| for the gradient scale
        amax = torch.empty(1, dtype=torch.float32, device=pred.device) if self._dyn_gscale else None
        loss, dpred = T.mse_loss_masked(pred.reshape(n, 7), _f32c(noise).reshape(n, 7), _f32c(part_valids).reshape(n), _u8(ref_part).reshape(n),
                                        amax=amax)
        self.backward(ctx, dpred, amax=amax)
        return loss


def _masked_stream(device, pct: int, from_top: bool = False):
    """experiment (PFPP_ENC_CU_FRACTION_PCT): a HIP stream restricted to the first pct % of every XCD's CUs (hipExtStreamCreateWithCUMask),
    wrapped for torch — the encoder then cannot take the whole chip from the transformer's dependency chain.  None when pct is 0."""
    if pct <= 0 or pct >= 100:
        return None
    import ctypes

    try:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipExtStreamCreateWithCUMask
    except (OSError, AttributeError):                 # another HIP runtime layout: fall back to an ordinary stream
        return None
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count          # 256: CU i lives on XCD i % 8
    if n_cu % 8 or n_cu < 64:
        return None
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    per_xcd = n_cu // 8
    keep = max(1, per_xcd * pct // 100)
    by_xcd = os.environ.get("PFPP_CU_MASK_MODE", "slots") == "xcd"      # experiment: whole XCDs instead of the same slots of every XCD
    for cu in range(n_cu):
        slot = cu // 8                                # CU index -> (slot = cu // 8, XCD = cu % 8)
        take = (cu % 8) < max(1, 8 * pct // 100) if by_xcd else ((per_xcd - 1 - slot) < keep if from_top else slot < keep)
        if take:
            mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0 or not st.value:
        return None
    _STREAM_CUS[int(st.value)] = sum(bin(int(w)).count("1") for w in mask)
    return torch.cuda.ExternalStream(st.value, device=device)


_STREAM_CUS = {}          # HIP stream handle -> number of CUs its mask leaves it (persistent kernels size their grid by it)


class FeaturePipeline:
    """Runs the frozen encoder of the NEXT training batch on its own HIP stream while the transformer forward / backward
    / optimizer of the current batch occupy the main stream.

    The encoder does not depend on the weights being trained (train_denoiser.py:33-35 freezes it), only on the batch
    and its freshly drawn (noise, timestep) — so step i+1's `add_noise -> rotate -> encode` can be issued before step i's
    transformer work.  The token-sized transformer GEMMs leave most of the 256 CUs idle; the encoder's big GEMMs fill
    them.  Order of the encoder's own state (BatchNorm running statistics) is preserved: all encoder work is on one stream.
    """

    def __init__(self, denoiser_module, device):
        self.model = denoiser_module                  # puzzlefusion_plusplus...Denoiser (encoder + noise_scheduler)
        self.device = device
        self.stream = None                            # chosen at the first issue (see _pick_stream)
        self.pending = None

    def _pick_stream(self):
        """the encoder's stream, once: CU-masked (PFPP_ENC_CU_FRACTION_PCT % of every XCD's CUs) when the caller's loop runs on a stream
        of its own.  hipExtStreamCreateWithCUMask makes a BLOCKING stream — it synchronises implicitly with the legacy default
        stream — so a loop that runs on the default stream gets an ordinary non-blocking stream instead (measured with the mask and the
        loop on the default stream: 11.7 ms instead of 8.0)."""
        on_default = torch.cuda.current_stream(self.device) == torch.cuda.default_stream(self.device)
        st = None if on_default else _masked_stream(self.device, int(os.environ.get("PFPP_ENC_CU_FRACTION_PCT", "50")))
        return st or torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PFPP_SIDE_PRIORITY", "0")))

    def _issue(self, data, gt, ref, noise, t):
        if self.stream is None:
            self.stream = self._pick_stream()
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)                 # inputs were produced on the main stream
        prev_wgs = ops.PERSISTENT_WGS
        if os.environ.get("PFPP_ENC_WGS_AUTO", "1") == "1":
            ops.PERSISTENT_WGS = _STREAM_CUS.get(int(self.stream.cuda_stream))   # one persistent workgroup per CU the stream may use
        try:
            with torch.cuda.stream(self.stream), torch.no_grad():
                noisy = self.model.noise_scheduler.add_noise(gt, noise, t)
                noisy = torch.where(ref.bool().unsqueeze(-1), gt, noisy)     # noisy[ref] = gt[ref] without the host sync of mask indexing
                latent, xyz = self.model._extract_features(data["part_pcs"], data["part_valids"], noisy)
        finally:
            ops.PERSISTENT_WGS = prev_wgs
        for v in (gt, ref, noise, t):
            # read on the encoder stream, allocated on the caller's: a caller that drops them right away (TrainingSchedule._prepare's
            # locals) must not get the blocks recycled under the encoder stream's pending reads
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return dict(noisy=noisy, latent=latent, xyz=xyz, noise=noise, t=t, event=ev)

    def take(self, data, gt, ref, draw):
        """-> features of the batch issued last (issued now when there is none); the caller issues the following one itself
        (issue_next) — at a point of its choice inside the iteration"""
        if self.pending is None:
            self.pending = self._issue(data, gt, ref, *draw())
        cur, self.pending = self.pending, None
        main = torch.cuda.current_stream()
        main.wait_event(cur["event"])
        for k in ("noisy", "latent", "xyz"):
            cur[k].record_stream(main)
        return cur

    def issue_next(self, data, gt, ref, draw) -> None:
        self.pending = self._issue(data, gt, ref, *draw())

    def next(self, data, gt, ref, draw):
        """-> features of the batch issued on the previous call (or now, the first time), and issues the following one.
        `draw()` returns (noise, timesteps) for a batch."""
        if self.pending is None:
            self.pending = self._issue(data, gt, ref, *draw())
        cur, self.pending = self.pending, self._issue(data, gt, ref, *draw())
        main = torch.cuda.current_stream()
        main.wait_event(cur["event"])
        for k in ("noisy", "latent", "xyz"):
            cur[k].record_stream(main)                # allocated on the encoder stream, consumed on the main stream
        return cur


class TrainingSchedule:
    """The benchmarked execution of the training iteration (bench.py TrainWorkload) behind the module surface:
    iterate the batches through this wrapper and run the usual loop body on what it yields —

        for batch in model.training_schedule(loader):          # or trainer.fit(model, train_dataloaders=model.training_schedule(loader))
            loss = model.training_step(batch, i); loss.backward(); optimizer.step(); optimizer.zero_grad()

    * the loop body runs on a HIGH-PRIORITY stream (the transformer's dependency chain sets the length of the iteration);
    * every batch is moved to the device with its valid-fragment layout derived on the host (no device read in the step);
    * the frozen encoder of batch i+1 (its own noise / timestep draw, add_noise, rotate, PointNet++/VQ encode) is issued on the
      CU-masked encoder stream BEFORE batch i is handed out, so it runs underneath batch i's transformer work; Denoiser.forward
      finds the result in batch["_pfpp_features"] and waits for it there (a consumer that prefetches one batch, like
      Lightning's data fetcher, therefore does not pull the encoder onto the critical path).
    Same arithmetic as the in-line path: the encoder depends on the batch and its random draw only (train_denoiser.py:33-35
    freezes it); only the ORDER of the BatchNorm running-statistics updates relative to the optimizer steps differs (none)."""

    def __init__(self, model, batches, device=None):
        self.model = model
        self.batches = batches
        self.device = torch.device(device) if device is not None else next(model.parameters()).device

    def __len__(self):
        return len(self.batches)

    def _prepare(self, batch, pipe):
        if batch is None:
            return None
        m, dev = self.model, self.device
        batch = m.on_before_batch_transfer(dict(batch))
        batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        batch = m.on_after_batch_transfer(batch)
        gt = torch.cat([batch["part_trans"], batch["part_rots"]], dim=-1).float().contiguous()
        noise = torch.randn(gt.shape, device=dev)
        t = torch.randint(0, m.noise_scheduler.config.num_train_timesteps, (gt.shape[0],), device=dev).long()
        batch["_pfpp_features"] = pipe._issue(batch, gt, batch["ref_part"], noise, t)
        return batch

    def __iter__(self):
        dev = self.device
        # one chain stream and one encoder pipeline per model: the caching allocator keeps a pool per stream, a fresh stream per epoch
        # would start every epoch with hipMalloc calls
        state = getattr(self.model, "_pfpp_schedule_state", None)
        if state is None or state[0] != dev:
            state = (dev, torch.cuda.Stream(device=dev, priority=-1), FeaturePipeline(self.model, dev))
            object.__setattr__(self.model, "_pfpp_schedule_state", state)
        _, chain, pipe = state
        it = iter(self.batches)
        outer = torch.cuda.current_stream(dev)
        chain.wait_stream(outer)
        try:
            with torch.cuda.stream(chain):
                nxt = self._prepare(next(it, None), pipe)
                while nxt is not None:
                    cur, nxt = nxt, self._prepare(next(it, None), pipe)
                    yield cur
        finally:
            outer.wait_stream(chain)


def take_features(batch):
    """the encoder results TrainingSchedule attached to a batch, made visible to the current stream (or None)"""
    f = batch.pop("_pfpp_features", None) if isinstance(batch, dict) else None
    if f is None:
        return None
    cur = torch.cuda.current_stream()
    cur.wait_event(f["event"])
    for k in ("noisy", "latent", "xyz"):
        f[k].record_stream(cur)               # allocated on the encoder stream, consumed here
    return f
