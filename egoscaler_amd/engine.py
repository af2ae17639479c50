"""Explicit forward / backward of the EgoScaler trajectory generator over libegomi.so kernels.

No tracing compiler and no torch autograd inside the path: every step below enqueues HIP kernels
through the C-ABI (egoscaler_amd.ops) on the current stream, with activations kept resident in HBM
buffers that are reused across steps (288 GB: nothing is recomputed, nothing is offloaded).

Reference call stack this replaces:
  pointllm/model/pointllm.py:90-178   PointLLMLlamaModel.forward  (encoder, projector, splice)
  pointbert/point_encoder.py:169-189  PointTransformer.forward
  HF modeling_llama.py:367-418        LlamaModel.forward (32 decoder layers)
  pointllm/model/pointllm.py:227-228  lm_head ;  train.py:174-184  loss, backward
"""
import os
from typing import Dict

import torch

from . import ops
from .config import EgoDims


class Workspace:
    """Named device buffers reused across steps (shape/dtype changes re-allocate)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, name, shape, dtype, zero=False):
        t = self.bufs.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or t.shape != shape or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs[name] = t
            if zero:
                t.zero_()
        elif zero:
            t.zero_()
        return t

    def bytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class Engine:
    def __init__(self, dims: EgoDims, params: Dict[str, torch.Tensor], device, dtype):
        self.dims, self.w, self.device, self.dtype = dims, params, device, dtype
        self.ws = Workspace(device)
        self.prepared = False
        self.prepare_epoch = 0         # bumped by prepare(): holders of derived weight copies (decode.Decoder caches) key on it
        self.lm_wT, self.lm_wT_stale, self.lm_wT_ver = None, True, None     # padded transpose of lm_head for its dgrad (backward_logits)
        self.param_ref = {}                # name -> nn.Parameter (set by the model shell): in-place updates by torch optimizers bump ITS
                                           # _version, not that of the `.data` aliases held in self.w
        self.grad_fresh = set()            # trainable weights whose main_grad must be overwritten, not accumulated, by their first wgrad
        self._xt_last = {}                 # wgrad workspace name -> identity of the activation it currently holds transposed
        self.main_grad: Dict[str, torch.Tensor] = {}     # fp32 gradient buffers of trainable tensors
        self.layer_flat: Dict[int, torch.Tensor] = {}    # decoder layer -> ONE flat fp32 block holding all of its gradients (= a DP bucket)
        self.trainable: Dict[str, bool] = {}
        self.ctx = None
        self.grad_sync = None          # optional dp.GradSync: notified when a gradient buffer is final
        self.pb_drop_override = None   # DropPath branch scales for the NEXT train-mode point-backbone pass instead of fresh draws (parity tests)
        self.reduced_grad = {}         # resident exchange (dp.GradSync(resident=True)): name -> bf16 view of the rank-summed gradient in the
        self.layer_offs = {}           # layer's wire buffer, read by EgoAdamW; layer_offs[l][name] = element offset inside the layer's flat block
        self._direct, self._direct_done = None, set()
        self.layer_final_hook = None   # EgoAdamW.arm(): called with l when decoder layer l's gradients are final (one rank, last backward of the step)
        self.param_events = {}         # EgoAdamW.step(overlap=True): group ("pre" | "embed" | layer index | "post") -> event of its side-stream update
        self.use_fused_attention = True
        self.use_fused_swiglu = os.environ.get("EGOMI_NO_FUSED_SWIGLU", "0") != "1"   # SwiGLU in the gate|up GEMM epilogue where the 256x256
                                                                                      # kernel runs (tests and A/B runs switch it off)
        self.use_tail_fuse = os.environ.get("EGOMI_NO_TAIL_FUSE", "0") != "1"         # K-sliced tail rows summed by the RMSNorm that reads them
        # trainable decoder weights: weight gradients and data gradients on the k-major 8-phase kernel (csrc/gemm_tn.hip) — no transposed
        # copies of dY / X per product, no resident W^T that must follow every optimizer step.  EGOMI_GEMM_TN=0: the round-1/2 route (A/B)
        self.use_tn = os.environ.get("EGOMI_GEMM_TN", "1") != "0"
        self.pb_trainer = None
        self.defer_splice_check = False     # hipGraph capture of a whole step: the marker verdict is copied to pinned memory by the graph and
        self.pending_splice = None          # looked at by the caller after the replay (check_pending_splice) instead of inside the forward pass
        self.pb_train_mode = False          # set by TrajPointLLMForCausalLM.train(): point backbone in train() mode
        self.prepared_bn_stale = False
        self.fold_stale = False

    @staticmethod
    def param_group_of(name):
        """The point of the forward pass at which a parameter is first read (EgoAdamW.step(overlap=True) updates in this order)."""
        if name.startswith("model.layers."):
            return int(name.split(".")[2])
        if name == "model.embed_tokens.weight":
            return "embed"
        if name in ("model.norm.weight", "lm_head.weight"):
            return "post"
        return "pre"                                       # projector, point backbone

    def wait_params(self, key):
        """Forward pass: the compute stream waits for the optimizer's side-stream update of this group, if one is still pending."""
        ev = self.param_events.pop(key, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wait_param_updates(self):
        """Everything the optimizer still has in flight (readers outside a forward pass: state_dict, generate, checkpoints)."""
        for key in list(self.param_events):
            self.wait_params(key)

    def _notify(self, name):
        if self.grad_sync is not None and name in self.trainable and name in self.main_grad:
            self.grad_sync.ready(name, self.main_grad[name])

    def _notify_layer(self, l):
        """A decoder layer's gradients are final: its flat block goes out as one bucket (bucketed per layer,
        reverse layer order, overlapped with the rest of backward).  Without an exchange (one rank) an armed optimizer may update the layer
        right away (EgoAdamW.arm: the update runs on its side stream under the backward of the layers below)."""
        one_rank = self.grad_sync is None or getattr(self.grad_sync, "local", False)
        hook = self.layer_final_hook if one_rank else None
        if self.grad_sync is None or l not in self.layer_flat:
            if hook is not None:
                hook(l)
            return
        if self._direct is not None:
            wire, views = self._direct
            for n, v in views.items():                                   # what no product wrote in wire precision (the two norm weights; a
                if n not in self._direct_done:                           # product the k-major kernel refused): cast from the fp32 block
                    if n in self.grad_fresh:
                        self.main_grad[n].zero_()
                        self.grad_fresh.discard(n)
                    ops.cast(self.main_grad[n].view(-1), wire.dtype, out=v.view(-1))
                self.reduced_grad[n] = v
            self._direct, self._direct_done = None, set()
            self.grad_sync.ready_resident(f"layer{l}", wire)
            if hook is not None:                                         # (local mode: the optimizer reads the wire buffer right away)
                hook(l)
            return
        self.grad_sync.ready_flat(f"layer{l}", self.layer_flat[l])
        if hook is not None:
            hook(l)
    def _begin_direct(self, l):
        """Resident exchange: this layer's weight gradients are produced in wire precision inside the layer's wire buffer (dp.py).  Only
        where every gradient of the layer is overwritten by this backward (no accumulation over micro-batches) on the bf16 k-major route."""
        self._direct, self._direct_done = None, set()
        gs = self.grad_sync
        if gs is None or not getattr(gs, "resident", False) or not self.use_tn or self.dtype != torch.bfloat16:
            return
        names = [n for n in self.layer_param_names(l) if n in self.trainable]
        if not names:
            return
        if l not in self.layer_flat:
            self._alloc_layer_grads(l)
        if any(self.w[n].dim() == 2 and n not in self.grad_fresh for n in names):
            return                                                       # accumulating: fp32 block + packing cast, as without resident mode
        wire = gs.resident_wire(f"layer{l}", self.layer_flat[l].numel(), self.device)
        if wire is None:
            return
        offs = self.layer_offs[l]
        self._direct = (wire, {n: wire[offs[n]:offs[n] + self.w[n].numel()].view(self.w[n].shape) for n in names})

    # ------------------------------------------------------------------------------------ setup
    def fold_batchnorm(self):
        """BatchNorm (eval) folded into the mini-PointNet convs (dvae.py:193-204 with running stats, model_arch.py:121-122).  Its own step:
        with a trainable point backbone the convs and the running statistics move every optimizer step, the LLM-side derived weights do not."""
        w, pb = self.w, self.dims.pb
        pre = "model.point_backbone.encoder."
        f = {}
        for conv, bn, key in (("first_conv.0", "first_conv.1", "c1"), ("second_conv.0", "second_conv.1", "c3")):
            W = w[pre + conv + ".weight"].float().squeeze(-1)
            b = w[pre + conv + ".bias"].float()
            g, beta = w[pre + bn + ".weight"].float(), w[pre + bn + ".bias"].float()
            mu, var = w[pre + bn + ".running_mean"].float(), w[pre + bn + ".running_var"].float()
            sc = g / torch.sqrt(var + pb.bn_eps)
            f[key + "_w"] = (W * sc[:, None]).to(self.dtype).contiguous()
            f[key + "_b"] = ((b - mu) * sc + beta).to(self.dtype).contiguous()
        f["c2_w"] = w[pre + "first_conv.3.weight"].squeeze(-1).contiguous()
        f["c4_w"] = w[pre + "second_conv.3.weight"].squeeze(-1).contiguous()
        self.folded = f
        self.fold_stale = False

    def prepare(self):
        """One-time derived weights: BatchNorm fold (fold_batchnorm), resident transposes / stacks of the decoder weights, RoPE tables."""
        self.wait_param_updates()                           # (derived weights read the parameters: nothing of an overlapped optimizer step may be in flight)
        w, pb, lm = self.w, self.dims.pb, self.dims.lm
        self.fold_batchnorm()
        # frozen decoder weights: keep W^T resident too, so dgrad (dX = dY.W) runs on the tuned
        # K-contiguous kernel instead of a transposing one (+13.5 GB at 7B; 288 GB HBM pays for it)
        self.wT = {}
        if self.dtype == torch.bfloat16:
            for l in range(lm.num_hidden_layers):
                for nm in self.layer_param_names(l):
                    if w[nm].dim() == 2 and not (self.use_tn and nm in self.trainable):
                        self.wT[nm] = ops.transpose(w[nm])     # trainable ones (EGOMI_GEMM_TN=0 only) are refreshed by after_weights_update()
        # [Wq;Wk;Wv] stacked: one N=3d product fills q|k|v (1102 vs 1010 TFLOP/s measured).  The model allocates the three side by side
        # (model_arch.py), so the stack is a VIEW of the parameters — also for trainable layers, whose stack thereby follows every
        # optimizer step for free; tensors that do not lie side by side (a .to() copy, foreign storage) are concatenated when frozen
        # and stay separate products when trainable
        self.wqkv, self.wgu_cat = {}, {}
        if self.dtype == torch.bfloat16:
            for l in range(lm.num_hidden_layers):
                p = f"model.layers.{l}.self_attn."
                names = [p + f"{n}_proj.weight" for n in "qkv"]
                v = self._side_by_side([w[n] for n in names])
                if v is not None:
                    self.wqkv[l] = v
                elif not any(n in self.trainable for n in names):
                    self.wqkv[l] = torch.cat([w[n] for n in names], 0)
                names = [f"model.layers.{l}.mlp.{n}_proj.weight" for n in ("gate", "up")]
                if self.use_tn and any(n in self.trainable for n in names):
                    v = self._side_by_side([w[n] for n in names])
                    if v is not None:
                        self.wgu_cat[l] = v                 # plain [Wgate;Wup]: gate columns, then up columns (no SwiGLU epilogue in training-all mode)
        # ... and [Wgate;Wup] for the forward, [WqT|WkT|WvT] / [WgateT|WupT] (K-concatenated) for dgrad: one long-K product
        # per block instead of 2-3 read-modify-write passes over dX; their separate W^T copies are dropped
        # [Wgate;Wup] is stacked with its rows INTERLEAVED in blocks of 32 (ffn % 32 == 0): gate|up then come out of the product in
        # the interleaved-32 layout (hidden unit c at column 64*(c/32) + c%32, up 32 further), in which one GEMM wave holds gate and
        # up of the same units, so silu(gate)*up is computed in the GEMM epilogue (EGOMI_EPI_SWIGLU) instead of a separate pass
        self.wgu, self.wqkvT, self.wguT = {}, {}, {}
        self.gu_il = lm.intermediate_size % 32 == 0
        if self.dtype == torch.bfloat16:
            for l in range(lm.num_hidden_layers):
                pa, pm = f"model.layers.{l}.self_attn.", f"model.layers.{l}.mlp."
                if l in self.wqkv and not any((pa + f"{n}_proj.weight") in self.trainable for n in "qkv"):
                    self.wqkvT[l] = torch.cat([self.wT.pop(pa + f"{n}_proj.weight") for n in "qkv"], 1)
                if not any((pm + f"{n}_proj.weight") in self.trainable for n in ("gate", "up")):
                    self.wgu[l] = self.stack_gate_up(w[pm + "gate_proj.weight"], w[pm + "up_proj.weight"])
                    self.wT.pop(pm + "gate_proj.weight"), self.wT.pop(pm + "up_proj.weight")
                    self.wguT[l] = ops.transpose(self.wgu[l])
        cos, sin = ops.rope_tables(lm.max_position_embeddings, lm.head_dim, lm.rope_theta)
        self.cos, self.sin = cos.to(self.device), sin.to(self.device)
        self.prepared = True
        self.prepare_epoch += 1

    @staticmethod
    def _side_by_side(ts):
        """One [sum rows, cols] view over 2-D tensors that lie back to back in one allocation, else None."""
        t0 = ts[0]
        if any(t.dim() != 2 or not t.is_contiguous() or t.shape[1] != t0.shape[1] or t.dtype != t0.dtype for t in ts):
            return None
        end = t0.data_ptr()
        for t in ts:
            if t.data_ptr() != end or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr():
                return None
            end += t.numel() * t.element_size()
        return torch.as_strided(t0, (sum(t.shape[0] for t in ts), t0.shape[1]), (t0.shape[1], 1))

    def stack_gate_up(self, gate, up):
        """[Wgate;Wup] -> [2*ffn, d]; rows interleaved in blocks of 32 when ffn allows (self.gu_il), plain concatenation otherwise."""
        if not self.gu_il:
            return torch.cat([gate, up], 0)
        Fd = gate.shape[0]
        return torch.stack([gate.view(Fd // 32, 32, -1), up.view(Fd // 32, 32, -1)], 1).reshape(2 * Fd, -1).contiguous()

    def after_weights_update(self):
        """Called by the optimizer after a step: the resident W^T of TRAINABLE decoder weights must
        follow the new values (a 2-byte transpose pass per weight; frozen ones never change)."""
        self.lm_wT_stale = True
        if not self.prepared:
            return
        if self.pb_trainable:
            self.fold_stale = True              # BN-folded convs of the eval path follow the new backbone weights: re-folded when that path next runs
                                                # (round 2 re-ran ALL of prepare() here: 13.5 GB of decoder-weight transposes and stacks per step, `bench.py --mode pc`)
        for nm, wt in self.wT.items():
            if nm in self.trainable:
                ops.transpose(self.w[nm], out=wt)

    def set_trainable(self, names):
        self.trainable = {n: True for n in names}

    def grad_buffer(self, name):
        g = self.main_grad.get(name)
        if g is None:
            if name.startswith("model.layers."):
                self._alloc_layer_grads(int(name.split(".")[2]))
                return self.main_grad[name]
            g = torch.zeros(self.w[name].shape, dtype=torch.float32, device=self.device)
            self.main_grad[name] = g
        return g

    def _alloc_layer_grads(self, l):
        """All trainable tensors of decoder layer l share one flat fp32 block (each starts on a 256-B boundary)."""
        names = [n for n in self.layer_param_names(l) if n in self.trainable]
        offs, total = [], 0
        for n in names:
            offs.append(total)
            total += (self.w[n].numel() + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.layer_flat[l] = flat
        self.layer_offs[l] = dict(zip(names, offs))
        for n, o in zip(names, offs):
            self.main_grad[n] = flat[o:o + self.w[n].numel()].view(self.w[n].shape)

    def zero_grad(self):
        # EgoAdamW.step(overlap=True) may still be reading these buffers on its side stream (the reference loop calls zero_grad right
        # after step, train.py:159): the compute stream waits for the updates in flight before it clears anything
        self.wait_param_updates()
        for g in self.main_grad.values():
            g.zero_()
        self.grad_fresh.clear()

    def lazy_zero_ok(self, name):
        """2-D weights whose whole gradient is written by _wgrad's product (decoder layers, lm_head, projector): their
        buffers may skip the zero pass and be overwritten by the first product of the step."""
        return self.w[name].dim() == 2 and (name.startswith("model.layers.") or name == "lm_head.weight" or name.startswith("model.point_proj."))

    def flush_fresh(self):
        """End of a backward: a buffer that no product touched this step must still read as zero."""
        for n in list(self.grad_fresh):
            self.main_grad[n].zero_()
        self.grad_fresh.clear()

    # ------------------------------------------------------------------------------------ pieces
    def _attention(self, qkv, B, S, H, hd, out, causal, key_mask, scale, keep_P):
        """qkv [B*S, 3*H*hd]; out [B*S, H*hd].  Unfused: scores (fp32) -> softmax -> P.V, all on the
        batched MFMA GEMM; no transposes, heads addressed by strides."""
        d = H * hd
        T = self.dtype
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        sc = self.ws.get("att_scores", (B * H, S, S), torch.float32)
        ops.gemm_raw(q, k, sc, S, S, hd, 3 * d, 3 * d, S, 0, 0, alpha=scale, batch=B * H, batch_inner=H,
                     strides=(S * 3 * d, hd, S * 3 * d, hd, H * S * S, S * S))
        Pm = torch.empty(B * H, S, S, dtype=T, device=self.device) if keep_P else self.ws.get("att_P", (B * H, S, S), T)
        ops.softmax(sc, B * H, H, S, S, Pm, causal=causal, key_mask=key_mask)
        ops.gemm_raw(Pm, v, out, S, hd, S, S, 3 * d, d, 0, 1, batch=B * H, batch_inner=H,
                     strides=(H * S * S, S * S, S * 3 * d, hd, S * d, hd))
        return Pm

    def _attention_bwd(self, qkv, Pm, d_out, dqkv, B, S, H, hd, scale):
        d = H * hd
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        dP = self.ws.get("att_scores", (B * H, S, S), torch.float32)
        ops.gemm_raw(d_out, v, dP, S, S, hd, d, 3 * d, S, 0, 0, batch=B * H, batch_inner=H,
                     strides=(S * d, hd, S * 3 * d, hd, H * S * S, S * S))
        dS = self.ws.get("att_dS", (B * H, S, S), self.dtype)
        ops.softmax_bwd(Pm, dP, dS, B * H * S, S)
        st_ss = (H * S * S, S * S)
        ops.gemm_raw(dS, k, dq, S, hd, S, S, 3 * d, 3 * d, 0, 1, alpha=scale, batch=B * H, batch_inner=H,
                     strides=(*st_ss, S * 3 * d, hd, S * 3 * d, hd))
        ops.gemm_raw(dS, q, dk, S, hd, S, S, 3 * d, 3 * d, 1, 1, alpha=scale, batch=B * H, batch_inner=H,
                     strides=(*st_ss, S * 3 * d, hd, S * 3 * d, hd))
        ops.gemm_raw(Pm, d_out, dv, S, hd, S, S, d, 3 * d, 1, 1, batch=B * H, batch_inner=H,
                     strides=(*st_ss, S * d, hd, S * 3 * d, hd))

    # ------------------------------------------------------------------------------------ PointBERT
    @torch.no_grad()
    def pointnet(self, nb2d, BG, K):
        """mini-PointNet (A6, dvae.py:207-221) on [BG*K, C] neighbourhood rows -> [BG, encoder_dims].
        BatchNorm (running stats) is folded into the two convs that precede a ReLU."""
        if not self.prepared:
            self.prepare()
        if self.fold_stale:
            self.fold_batchnorm()
        w, pb, T, ws, f = self.w, self.dims.pb, self.dtype, self.ws, self.folded
        pre = "model.point_backbone."
        h1 = ops.linear_smallk(nb2d, f["c1_w"], f["c1_b"], act=ops.ACT_RELU, out=ws.get("pn_h1", (BG * K, pb.pn_c1), T))
        h2 = ops.mm(h1, f["c2_w"], out=ws.get("pn_h2", (BG * K, pb.pn_c2), T), bias=w[pre + "encoder.first_conv.3.bias"])
        cat = ops.group_max(h2, BG, K, pb.pn_c2, concat=True, out=ws.get("pn_cat", (BG * K, 2 * pb.pn_c2), T))
        h3 = ops.mm(cat, f["c3_w"], out=ws.get("pn_h3", (BG * K, pb.pn_c3), T), bias=f["c3_b"], act=ops.ACT_RELU)
        h4 = ops.mm(h3, f["c4_w"], out=ws.get("pn_h4", (BG * K, pb.encoder_dims), T), bias=w[pre + "encoder.second_conv.3.bias"])
        return ops.group_max(h4, BG, K, pb.encoder_dims, out=ws.get("pn_tok", (BG, pb.encoder_dims), T))

    @torch.no_grad()
    def point_backbone(self, pts: torch.Tensor, fps_start) -> torch.Tensor:
        """pts [B,N,C] f32 -> [B, G+1, D] (dtype T).  Frozen/eval path (model_arch.py:33-36,121-122)."""
        if not self.prepared:
            self.prepare()
        w, pb, T, ws = self.w, self.dims.pb, self.dtype, self.ws
        pre = "model.point_backbone."
        B, N, C = pts.shape
        G, K, D, Pn = pb.num_group, pb.group_size, pb.trans_dim, pb.point_token_len
        idx, center = ops.fps(pts, fps_start, G)                                   # A3
        _, nb = ops.knn_group(pts, center, K, out_dtype=T)                         # A4+A5  [B,G,K,C]
        BG = B * G
        tok = self.pointnet(nb.view(BG * K, C), BG, K)                             # A6
        # tokens + positional embedding, written straight into rows 1..G of [B, G+1, D]
        x = ws.get("pb_x", (B, Pn, D), T)
        pos = ws.get("pb_pos", (B, Pn, D), T)
        x[:, 0] = w[pre + "cls_token"].view(1, D)
        pos[:, 0] = w[pre + "cls_pos"].view(1, D)
        ops.gemm_raw(tok, w[pre + "reduce_dim.weight"], x[:, 1:], G, D, pb.encoder_dims, pb.encoder_dims, pb.encoder_dims, D,
                     bias=w[pre + "reduce_dim.bias"], batch=B, strides=(G * pb.encoder_dims, 0, 0, 0, Pn * D, 0))
        ph = ops.linear_smallk(center.view(BG, 3), w[pre + "pos_embed.0.weight"], w[pre + "pos_embed.0.bias"], act=ops.ACT_GELU,
                               out=ws.get("pb_ph", (BG, pb.pos_hidden), T))
        ops.gemm_raw(ph, w[pre + "pos_embed.2.weight"], pos[:, 1:], G, D, pb.pos_hidden, pb.pos_hidden, pb.pos_hidden, D,
                     bias=w[pre + "pos_embed.2.bias"], batch=B, strides=(G * pb.pos_hidden, 0, 0, 0, Pn * D, 0))
        M = B * Pn
        H, hd = pb.num_heads, pb.head_dim
        xs = ws.get("pb_xs", (M, D), T)
        h = ws.get("pb_h", (M, D), T)
        qkv = ws.get("pb_qkv", (M, 3 * D), T)
        ao = ws.get("pb_ao", (M, D), T)
        x1 = ws.get("pb_x1", (M, D), T)
        mid = ws.get("pb_mid", (M, pb.mlp_ratio * D), T)
        xf = x.view(M, D)
        for i in range(pb.depth):
            p = f"{pre}blocks.blocks.{i}."
            ops.layernorm(xf, w[p + "norm1.weight"], w[p + "norm1.bias"], pb.ln_eps, add=pos.view(M, D), sum_out=xs, out=h)
            ops.mm(h, w[p + "attn.qkv.weight"], out=qkv)
            if self.use_fused_attention and T == torch.bfloat16 and hd == 64:
                ops.attn_fwd(qkv, B, Pn, H, hd, hd ** -0.5, ao, None, causal=False)       # scores never reach HBM
            else:
                self._attention(qkv, B, Pn, H, hd, ao, False, None, hd ** -0.5, False)
            ops.mm(ao, w[p + "attn.proj.weight"], out=x1, bias=w[p + "attn.proj.bias"], residual=xs)
            ops.layernorm(x1, w[p + "norm2.weight"], w[p + "norm2.bias"], pb.ln_eps, out=h)
            ops.mm(h, w[p + "mlp.fc1.weight"], out=mid, bias=w[p + "mlp.fc1.bias"], act=ops.ACT_GELU)
            ops.mm(mid, w[p + "mlp.fc2.weight"], out=xf, bias=w[p + "mlp.fc2.bias"], residual=x1)
        out = torch.empty(B, Pn, D, dtype=T, device=self.device)
        ops.layernorm(xf, w[pre + "norm.weight"], w[pre + "norm.bias"], pb.ln_eps, out=out.view(M, D))
        return out

    # ------------------------------------------------------------------------------------ forward
    def forward_hidden(self, input_ids, attention_mask, point_clouds, fps_start, save=True, kv_sink=None):
        """-> final-normed hidden [B*S, d]; fills self.ctx for backward when save=True.
        kv_sink(layer, qkv, B, S): optional prefill hook that receives the post-RoPE q|k|v buffer of every
        layer (decode.Decoder copies k, v into its static cache)."""
        if not self.prepared:
            self.prepare()
        w, dims, T, ws = self.w, self.dims, self.dtype, self.ws
        lm, pb, tok = dims.lm, dims.pb, dims.tok
        if input_ids.dtype != torch.int64 or not input_ids.is_contiguous():
            input_ids = input_ids.to(torch.int64).contiguous()      # kernels index ids as a dense [B,S] int64 array
        B, S = input_ids.shape
        d, Fd, H, hd, L = lm.hidden_size, lm.intermediate_size, lm.num_attention_heads, lm.head_dim, lm.num_hidden_layers
        M = B * S
        Pn = pb.point_token_len
        ctx = {"B": B, "S": S, "ids": input_ids, "layers": []} if save else None
        past = 0
        self.wait_params("pre")
        # ---- point branch (pointllm.py:112-129): only when S != 1 (prefill / training)
        feats_proj, start_pos, pending_err, cloud_idx, Bc = None, None, None, None, 0
        if point_clouds is not None and S != 1:
            if isinstance(point_clouds, (list, tuple)):                          # pointllm.py:117-122
                fl = [self.point_backbone(pc[None].to(self.device, torch.float32), [int(fps_start[i])]) for i, pc in enumerate(point_clouds)]
                feats = torch.cat(fl, 0)
            elif save and self.pb_train_mode and self.pb_trainable:
                # --unfreeze_pc_encoder in train(): batch-statistics BatchNorm, DropPath, activations kept for backward
                from .pointbert_train import PointBackboneTrainer
                if self.pb_trainer is None:
                    self.pb_trainer = PointBackboneTrainer(self)
                drop = self.pb_drop_override if self.pb_drop_override is not None else self.pb_trainer.drop_scales(point_clouds.shape[0])
                self.pb_drop_override = None                # one forward pass only (tests hand in recorded DropPath draws: [depth, 2, B] scales)
                feats, pb_ctx = self.pb_trainer.forward(point_clouds.to(self.device, torch.float32), fps_start, drop)
                ctx["pb_ctx"] = pb_ctx
                self.prepared_bn_stale = True
            else:
                feats = self.point_backbone(point_clouds.to(self.device, torch.float32), fps_start)
            Bc = feats.shape[0]                      # clouds given: one per sample in EgoScaler's data; the splice follows the reference's running
            fm = feats.reshape(Bc * Pn, pb.trans_dim)   # cloud index, so a batch whose samples hold several segments may bring more (pointllm.py:135-156)
            acts = [fm]
            nh = len(pb.projection_hidden_dim)
            cur = fm
            for j in range(nh):                                                    # pointllm.py:67-81
                Wj, bj = w[f"model.point_proj.{2 * j}.weight"], w[f"model.point_proj.{2 * j}.bias"]
                pre_act = ops.mm(cur, Wj, bias=bj, out=ws.get(f"pp_pre{j}", (Bc * Pn, Wj.shape[0]), T))
                cur = ops.gelu(pre_act, out=ws.get(f"pp_act{j}", (Bc * Pn, Wj.shape[0]), T))
                acts += [pre_act, cur]
            Wl, bl = w[f"model.point_proj.{2 * nh}.weight"], w[f"model.point_proj.{2 * nh}.bias"]
            feats_proj = ops.mm(cur, Wl, bias=bl, out=ws.get("pp_out", (Bc * Pn, d), T))
            sp, err, cloud_idx = ops.splice_scan(input_ids, tok, Pn, Bc)
            # the reference checks the markers on the host right here (pointllm.py:137-151) and stalls the stream for it.  The
            # scan kernel already answers an inconsistent sample with start_pos = -1 (= text only: nothing downstream reads out
            # of range), so the verdict is copied to pinned memory now and looked at when the rest of the forward pass has been
            # queued — the same exceptions, raised by the same call, without 0.6 ms of idle GPU per step
            pending_err = self._stash_splice_err(err)
            start_pos = sp
            if save:
                ctx["pp_acts"] = acts
        if save:
            ctx["start_pos"], ctx["cloud_idx"], ctx["n_clouds"] = start_pos, cloud_idx, Bc
            ctx["has_points"] = feats_proj is not None
        x = ws.get("x_emb", (B, S, d), T) if not save else torch.empty(B, S, d, dtype=T, device=self.device)
        self.wait_params("embed")
        ops.embed_splice(input_ids, w["model.embed_tokens.weight"], feats_proj, start_pos, Pn, out=x, cloud_idx=cloud_idx)
        x = x.view(M, d)
        key_mask = None
        if attention_mask is not None:
            key_mask = attention_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        scale = hd ** -0.5
        fused = self.use_fused_attention and T == torch.bfloat16 and hd == 128
        # the K-sliced tail rows of o_proj / down_proj (and, in backward, of the qkv / gate|up dgrads) are summed by the RMSNorm
        # kernel that reads them instead of by a combine pass (EGOMI_EPI_SLABS, include/egomi.h).  EGOMI_NO_TAIL_FUSE=1: A/B switch
        defer = self.use_tail_fuse and T == torch.bfloat16
        pend = pend_res = None
        fuse_swiglu = self.use_fused_swiglu and T == torch.bfloat16 and self.gu_il and (2 * Fd) % 256 == 0 and ops.gemm_kernel_id(M, 2 * Fd, d) == 2
        for l in range(L):
            p = f"model.layers.{l}."
            self.wait_params(l)
            if save:
                lc = {"x_in": x, "rstd1": torch.empty(M, dtype=torch.float32, device=self.device),
                      "rstd2": torch.empty(M, dtype=torch.float32, device=self.device),
                      "qkv": torch.empty(M, 3 * d, dtype=T, device=self.device),
                      "gu": torch.empty(M, 2 * Fd, dtype=T, device=self.device)}
                keep_in = self.any_layer_trainable
                h = torch.empty(M, d, dtype=T, device=self.device) if keep_in else ws.get("h", (M, d), T)
                ao = torch.empty(M, d, dtype=T, device=self.device) if (keep_in or fused) else ws.get("ao", (M, d), T)
                h2 = torch.empty(M, d, dtype=T, device=self.device) if keep_in else ws.get("h2", (M, d), T)
                act = torch.empty(M, Fd, dtype=T, device=self.device) if keep_in else ws.get("act", (M, Fd), T)
                x_mid = torch.empty(M, d, dtype=T, device=self.device)
                x_out = torch.empty(M, d, dtype=T, device=self.device)
                qkv, gu, rstd1, rstd2 = lc["qkv"], lc["gu"], lc["rstd1"], lc["rstd2"]
            else:
                h, ao, h2, act = ws.get("h", (M, d), T), ws.get("ao", (M, d), T), ws.get("h2", (M, d), T), ws.get("act", (M, Fd), T)
                x_mid, x_out = ws.get("x_mid", (M, d), T), ws.get(f"x_out{l & 1}", (M, d), T)
                qkv, gu, rstd1, rstd2 = ws.get("qkv", (M, 3 * d), T), ws.get("gu", (M, 2 * Fd), T), None, None
            ops.rmsnorm(x, w[p + "input_layernorm.weight"], lm.rms_norm_eps, rstd=rstd1, out=h, tail=pend, tail_residual=pend_res)
            pend = pend_res = None
            t_qkv = None
            if l in self.wqkv:
                if defer:
                    _, t_qkv = ops.mm(h, self.wqkv[l], out=qkv, defer_tail=True)
                else:
                    ops.mm(h, self.wqkv[l], out=qkv)
            else:
                ops.mm(h, w[p + "self_attn.q_proj.weight"], out=qkv[:, :d])
                ops.mm(h, w[p + "self_attn.k_proj.weight"], out=qkv[:, d:2 * d])
                ops.mm(h, w[p + "self_attn.v_proj.weight"], out=qkv[:, 2 * d:])
            if t_qkv is not None:
                ops.rope_qkv_tail_(qkv, self.cos, self.sin, M, S, past, H, hd, 3 * d, t_qkv)
            else:
                ops.rope_(qkv, self.cos, self.sin, M, S, past, 2 * H, hd, 3 * d)      # q and k heads are adjacent columns
            if kv_sink is not None:
                kv_sink(l, qkv, B, S)
            lse = None
            if fused:
                # fused flash-style kernel: scores never reach HBM; LSE (and the output) are kept for backward
                Pm = None
                lse = torch.empty(B, H, S, dtype=torch.float32, device=self.device) if save else ws.get("att_lse", (B, H, S), torch.float32)
                ops.attn_fwd(qkv, B, S, H, hd, scale, ao, lse, causal=True, key_mask=key_mask)
            else:
                Pm = self._attention(qkv, B, S, H, hd, ao, True, key_mask, scale, save)
            if defer:                                          # the K-sliced tail rows of the product are summed by the norm that reads them
                _, t_o = ops.mm(ao, w[p + "self_attn.o_proj.weight"], out=x_mid, residual=x, defer_tail=True)
                ops.rmsnorm(x_mid, w[p + "post_attention_layernorm.weight"], lm.rms_norm_eps, rstd=rstd2, out=h2, tail=t_o, tail_residual=x)
            else:
                ops.mm(ao, w[p + "self_attn.o_proj.weight"], out=x_mid, residual=x)
                ops.rmsnorm(x_mid, w[p + "post_attention_layernorm.weight"], lm.rms_norm_eps, rstd=rstd2, out=h2)
            if l in self.wgu and self.gu_il:
                if fuse_swiglu:
                    ops.mm(h2, self.wgu[l], out=gu, swiglu_out=act)          # act leaves the GEMM epilogue; gu (interleaved-32) is kept for backward
                else:
                    ops.mm(h2, self.wgu[l], out=gu)
                    ops.swiglu_il(gu, act)
            else:
                if l in self.wgu or l in self.wgu_cat:
                    ops.mm(h2, self.wgu[l] if l in self.wgu else self.wgu_cat[l], out=gu)
                else:
                    ops.mm(h2, w[p + "mlp.gate_proj.weight"], out=gu[:, :Fd])
                    ops.mm(h2, w[p + "mlp.up_proj.weight"], out=gu[:, Fd:])
                ops.swiglu(gu[:, :Fd], gu[:, Fd:], act)
            if defer:                                          # ... here by the NEXT layer's input norm (or the final norm)
                _, pend = ops.mm(act, w[p + "mlp.down_proj.weight"], out=x_out, residual=x_mid, defer_tail=True)
                pend_res = x_mid
            else:
                ops.mm(act, w[p + "mlp.down_proj.weight"], out=x_out, residual=x_mid)
            if save:
                lc.update(P=Pm, lse=lse, x_mid=x_mid, h=h, ao=ao, h2=h2, act=act)
                ctx["layers"].append(lc)
            x = x_out
        self.wait_params("post")
        rstd_f = torch.empty(M, dtype=torch.float32, device=self.device) if save else None
        hn = torch.empty(M, d, dtype=T, device=self.device)
        ops.rmsnorm(x, w["model.norm.weight"], lm.rms_norm_eps, rstd=rstd_f, out=hn, tail=pend, tail_residual=pend_res)
        if save:
            ctx.update(x_last=x, rstd_f=rstd_f, hn=hn, key_mask=key_mask)
            self.ctx = ctx
        if self.defer_splice_check:
            self.pending_splice = pending_err
        else:
            self._check_splice_err(pending_err)
        return hn

    def _stash_splice_err(self, err):
        host = self._splice_host.get(err.numel()) if hasattr(self, "_splice_host") else None
        if host is None:
            if not hasattr(self, "_splice_host"):
                self._splice_host = {}
            host = self._splice_host[err.numel()] = torch.empty(err.numel(), dtype=torch.int32).pin_memory() if err.is_cuda else torch.empty(err.numel(), dtype=torch.int32)
        host.copy_(err, non_blocking=True)
        ev = None
        if err.is_cuda and not self.defer_splice_check:
            ev = torch.cuda.Event()
            ev.record()
        return host, ev

    def check_pending_splice(self):
        """defer_splice_check mode: raise what the forward pass would have raised (call after the stream has been synchronised)."""
        p, self.pending_splice = self.pending_splice, None
        self._check_splice_err(p)

    @staticmethod
    def _check_splice_err(pending):
        if pending is None:
            return
        e, ev = pending
        if ev is not None:
            ev.synchronize()
        if int(e.max()) != 0:
            code = int(e[e != 0][0])
            if code == 1:
                raise ValueError("The number of point start tokens and point end tokens should be the same.")
            if code == 2:
                raise ValueError("The point end token should follow the point start token.")
            # pointllm.py:143 `point_features[cur_point_idx]`: the running cloud index (one per text-only sample, one per SEGMENT of the others) ran past the clouds
            raise IndexError("index out of range: the batch's point segments need more point clouds than were given (pointllm.py:143,156)")

    def logits(self, hn, rows=None, padded=False):
        """lm_head (pointllm.py:227-228).  hn [M,d] -> [M,V].  padded=True (training step): the result is a view of a
        [M, V64] buffer (V rounded up to 64, pad columns zero) so that the lm_head dgrad can run K-contiguous on the
        tuned kernel with K = V64 (V = 32003 + new tokens is odd: rows of a dense [M,V] array are not even 4-B aligned)."""
        W = self.w["lm_head.weight"]
        M, V = hn.shape[0], W.shape[0]
        if padded and self.dtype == torch.bfloat16:
            Vp = (V + 63) // 64 * 64
            buf = torch.empty(M, Vp, dtype=self.dtype, device=self.device)
            buf[:, V:].zero_()
            return ops.mm(hn, W, out=buf[:, :V])
        return ops.mm(hn, W, out=torch.empty(M, V, dtype=self.dtype, device=self.device))

    @property
    def pb_trainable(self):
        return any(n.startswith("model.point_backbone.") for n in self.trainable)

    @property
    def any_layer_trainable(self):
        return any(n.startswith("model.layers.") for n in self.trainable)

    # ------------------------------------------------------------------------------------ backward
    def _dgrad(self, dY, name, out, residual=None):
        """out = dY . W (+ residual).  Uses the resident W^T (tuned NT kernel) for frozen weights."""
        wt = self.wT.get(name) if self.prepared else None
        if wt is not None:
            return ops.mm(dY, wt, out=out, residual=residual)
        return self._dgrad_w(dY, self.w[name], out, residual)

    def _dgrad_w(self, dY, W, out, residual=None):
        """out = dY . W (+ residual) with W [N, K] as it lies in memory."""
        if self.use_tn and self.dtype == torch.bfloat16 and (residual is None or residual is out):
            acc = residual is not None                              # "+ residual" with residual == out is an accumulation into out
            if ops.mm_kernel_id(dY, W, out, b_layout=1, accumulate=acc) == 3:
                return ops.mm(dY, W, out=out, b_layout=1, accumulate=acc)         # W [N, K] read k-major: no W^T copy
        return ops.mm(dY, W, out=out, b_layout=1, residual=residual)

    def _wgrad(self, name, dY, X):
        """main_grad[name] (fp32 [N,K]) += dY^T [N,M] . X [M,K].  bf16: both operands are transposed
        into zero-padded [*, M64] buffers so the product runs K-contiguous on the tuned kernel."""
        if name not in self.trainable:
            return
        if self._direct is not None and name in self._direct[1] and name in self.grad_fresh:
            v = self._direct[1][name]
            if ops.mm_kernel_id(dY, X, v, a_layout=1, b_layout=1) == 3:
                ops.mm(dY, X, out=v, a_layout=1, b_layout=1)             # bf16 result straight into the layer's wire buffer
                self.grad_fresh.discard(name)
                self._direct_done.add(name)
                return
        g = self.grad_buffer(name)
        # a buffer still marked fresh has not been zeroed this step (lazy_zero_names): the first product overwrites it,
        # which saves the zero pass and the read of C (26 GB each per step when every layer is trained)
        acc = name not in self.grad_fresh
        self.grad_fresh.discard(name)
        self._wgrad_into(g, acc, dY, X)

    def _wgrad_stacked(self, names, dY, X):
        """The gradients of weights that share their input X (q|k|v, gate|up), dY [M, sum N_i] holding their output gradients side
        by side: ONE product into the stacked view of their gradient buffers when those lie back to back in the layer's flat block
        and agree on overwrite-vs-accumulate; the separate products otherwise."""
        if self._direct is not None and all(n in self._direct[1] and n in self.grad_fresh for n in names):
            g = self._side_by_side([self._direct[1][n] for n in names])
            if g is not None and ops.mm_kernel_id(dY, X, g, a_layout=1, b_layout=1) == 3:
                ops.mm(dY, X, out=g, a_layout=1, b_layout=1)
                for n in names:
                    self.grad_fresh.discard(n)
                    self._direct_done.add(n)
                return
        if self.use_tn and self.dtype == torch.bfloat16 and all(n in self.trainable for n in names):
            gs = [self.grad_buffer(n) for n in names]
            fresh = [n in self.grad_fresh for n in names]
            g = self._side_by_side(gs) if all(f == fresh[0] for f in fresh) else None
            if g is not None and ops.mm_kernel_id(dY, X, g, a_layout=1, b_layout=1, accumulate=not fresh[0]) == 3:
                for n in names:
                    self.grad_fresh.discard(n)
                ops.mm(dY, X, out=g, a_layout=1, b_layout=1, accumulate=not fresh[0])
                return
        c = 0
        for n in names:
            N = self.w[n].shape[0]
            self._wgrad(n, dY[:, c:c + N], X)
            c += N

    def _wgrad_into(self, g, acc, dY, X):
        Mr, N = dY.shape
        K = X.shape[1]
        if self.use_tn and self.dtype == torch.bfloat16 and ops.mm_kernel_id(dY, X, g, a_layout=1, b_layout=1, accumulate=acc) == 3:
            ops.mm(dY, X, out=g, a_layout=1, b_layout=1, accumulate=acc)          # dY^T . X with both operands as they lie in memory
        elif self.dtype == torch.bfloat16 and Mr >= 128 and N * K >= 128 * 128:
            Mp = (Mr + 63) // 64 * 64
            dYt = ops.transpose(dY, ldo=Mp, out=self.ws.get(f"wg_dYt_{N}_{Mp}", (N, Mp), self.dtype))
            xname = f"wg_Xt_{K}_{Mp}"
            Xt = self.ws.get(xname, (K, Mp), self.dtype)
            xkey = (X.data_ptr(), X.stride(0), Mr)
            if self._xt_last.get(xname) != xkey:                  # q, k, v (and gate, up) share their input: transposed once
                ops.transpose(X, ldo=Mp, out=Xt)
                self._xt_last[xname] = xkey
            ops.mm(dYt, Xt, out=g, accumulate=acc)
        else:
            ops.mm(dY, X, out=g, a_layout=1, b_layout=1, accumulate=acc)

    def _bgrad(self, name, dY):
        if name in self.trainable:
            ops.colsum_(dY, self.grad_buffer(name))

    def backward_hidden(self, d_hn):
        """d_hn: gradient w.r.t. the final-normed hidden [M,d].  Accumulates fp32 main_grad of the
        trainable tensors (model_arch.py:33-51 decides which)."""
        ctx, w, dims, T, ws = self.ctx, self.w, self.dims, self.dtype, self.ws
        lm, pb = dims.lm, dims.pb
        B, S = ctx["B"], ctx["S"]
        d, Fd, H, hd, L = lm.hidden_size, lm.intermediate_size, lm.num_attention_heads, lm.head_dim, lm.num_hidden_layers
        M = B * S
        scale = hd ** -0.5
        tr = self.trainable
        dw = self.grad_buffer("model.norm.weight") if "model.norm.weight" in tr else None
        dx = ops.rmsnorm_bwd(d_hn, ctx["x_last"], w["model.norm.weight"], ctx["rstd_f"], dw=dw, out=ws.get("dx_a", (M, d), T))
        # frozen layers only: with trainable layers the wgrad products between a dgrad and its norm would reuse the slab area
        defer_b = self.use_tail_fuse and T == torch.bfloat16 and not self.any_layer_trainable
        fuse_swiglu_bwd = self.use_fused_swiglu and T == torch.bfloat16 and self.gu_il and Fd % 64 == 0 and ops.gemm_kernel_id(M, Fd, d) == 2
        for l in reversed(range(L)):
            p = f"model.layers.{l}."
            lc = ctx["layers"][l]
            gu, qkv = lc["gu"], lc["qkv"]
            self._begin_direct(l)
            # ---- MLP
            dgu = ws.get("dgu", (M, 2 * Fd), T)
            wt_down = self.wT.get(p + "mlp.down_proj.weight") if self.prepared else None
            if fuse_swiglu_bwd and wt_down is not None and l in self.wguT:
                # d(act) = dx . W_down never reaches memory: the GEMM epilogue reads gate|up and writes d(gate|up) (EGOMI_EPI_SWIGLU_BWD)
                ops.mm(dx, wt_down, out=dgu, swiglu_bwd_gu=gu)
            else:
                d_act = self._dgrad(dx, p + "mlp.down_proj.weight", ws.get("d_act", (M, Fd), T))
                if self.prepared and l in self.wguT and self.gu_il:
                    ops.swiglu_il_bwd(d_act, gu, dgu)                           # gu / dgu in the interleaved-32 layout of the stacked weight
                else:
                    ops.swiglu_bwd(d_act, gu[:, :Fd], gu[:, Fd:], dgu[:, :Fd], dgu[:, Fd:])
            self._wgrad(p + "mlp.down_proj.weight", dx, lc["act"])
            t_h2 = None
            if self.prepared and l in self.wguT:
                if defer_b:
                    d_h2, t_h2 = ops.mm(dgu, self.wguT[l], out=ws.get("d_h", (M, d), T), defer_tail=True)
                else:
                    d_h2 = ops.mm(dgu, self.wguT[l], out=ws.get("d_h", (M, d), T))
            elif self.prepared and l in self.wgu_cat:
                d_h2 = self._dgrad_w(dgu, self.wgu_cat[l], ws.get("d_h", (M, d), T))          # one K = 2*ffn product over [Wgate;Wup] in place
            else:
                d_h2 = self._dgrad(dgu[:, :Fd], p + "mlp.gate_proj.weight", ws.get("d_h", (M, d), T))
                self._dgrad(dgu[:, Fd:], p + "mlp.up_proj.weight", d_h2, residual=d_h2)
            self._wgrad_stacked([p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"], dgu, lc["h2"])
            n2 = p + "post_attention_layernorm.weight"
            d_mid = ops.rmsnorm_bwd(d_h2, lc["x_mid"], w[n2], lc["rstd2"], dx_add=dx,
                                    dw=self.grad_buffer(n2) if n2 in tr else None, out=ws.get("dx_b", (M, d), T), tail=t_h2)
            # ---- attention
            d_ao = self._dgrad(d_mid, p + "self_attn.o_proj.weight", ws.get("d_ao", (M, d), T))
            self._wgrad(p + "self_attn.o_proj.weight", d_mid, lc["ao"])
            dqkv = ws.get("dqkv", (M, 3 * d), T)
            if lc["lse"] is not None:
                # dq, dk leave the kernels already rotated back (the inverse RoPE pass is fused into their epilogues)
                ops.attn_bwd(qkv, lc["ao"], lc["lse"], d_ao, dqkv, ws.get("att_delta", (B, H, S), torch.float32), B, S, H, hd, scale,
                             causal=True, key_mask=ctx["key_mask"], rope=(self.cos, self.sin))
            else:
                self._attention_bwd(qkv, lc["P"], d_ao, dqkv, B, S, H, hd, scale)
                ops.rope_(dqkv, self.cos, self.sin, M, S, 0, 2 * H, hd, 3 * d, inverse=True)
            t_h = None
            if self.prepared and l in self.wqkvT:
                if defer_b:
                    d_h, t_h = ops.mm(dqkv, self.wqkvT[l], out=ws.get("d_h", (M, d), T), defer_tail=True)
                else:
                    d_h = ops.mm(dqkv, self.wqkvT[l], out=ws.get("d_h", (M, d), T))
            elif self.prepared and l in self.wqkv and self.use_tn and T == torch.bfloat16:
                d_h = self._dgrad_w(dqkv, self.wqkv[l], ws.get("d_h", (M, d), T))             # one K = 3d product over [Wq;Wk;Wv] in place
            else:
                d_h = self._dgrad(dqkv[:, :d], p + "self_attn.q_proj.weight", ws.get("d_h", (M, d), T))
                self._dgrad(dqkv[:, d:2 * d], p + "self_attn.k_proj.weight", d_h, residual=d_h)
                self._dgrad(dqkv[:, 2 * d:], p + "self_attn.v_proj.weight", d_h, residual=d_h)
            self._wgrad_stacked([p + f"self_attn.{nm}_proj.weight" for nm in "qkv"], dqkv, lc["h"])
            n1 = p + "input_layernorm.weight"
            dx = ops.rmsnorm_bwd(d_h, lc["x_in"], w[n1], lc["rstd1"], dx_add=d_mid,
                                 dw=self.grad_buffer(n1) if n1 in tr else None, out=ws.get("dx_a", (M, d), T), tail=t_h)
            self._notify_layer(l)
        # ---- embedding + splice + projector (pointllm.py:107,126-129,155)
        Pn = pb.point_token_len
        V = lm.vocab_size
        emb_name = "model.embed_tokens.weight"
        Bc = ctx["n_clouds"]
        d_feats = ws.get("d_pp_out", (Bc * Pn, d), T) if ctx["has_points"] else None
        if d_feats is not None:
            d_feats.zero_()                                    # a cloud no sample received (text-only sample, skipped index) has a zero gradient
        ops.embed_splice_bwd(dx.view(B, S, d), ctx["ids"], ctx["start_pos"], Pn, V,
                             self.grad_buffer(emb_name) if emb_name in tr else None, d_feats, cloud_idx=ctx["cloud_idx"])
        if ctx["has_points"]:
            acts = ctx["pp_acts"]
            nh = len(pb.projection_hidden_dim)
            g = d_feats
            for j in range(nh, -1, -1):
                Wn, bn = f"model.point_proj.{2 * j}.weight", f"model.point_proj.{2 * j}.bias"
                x_in = acts[2 * j]                     # input of linear j: feats (j=0) or gelu output
                self._wgrad(Wn, g, x_in)
                self._bgrad(bn, g)
                if j > 0:
                    if T == torch.bfloat16:
                        # dX = dY . W on the tuned K-contiguous kernel: the (trainable, hence changing) projector weight is
                        # transposed on the fly (a 16-MB pass) instead of running the transposing generic kernel (0.34 ms each)
                        wt = ops.transpose(w[Wn], out=ws.get(f"pp_wT{j}", (w[Wn].shape[1], w[Wn].shape[0]), T))
                        g_in = ops.mm(g, wt, out=ws.get(f"d_pp_act{j}", (Bc * Pn, w[Wn].shape[1]), T))
                    else:
                        g_in = ops.mm(g, w[Wn], out=ws.get(f"d_pp_act{j}", (Bc * Pn, w[Wn].shape[1]), T), b_layout=1)
                    g = ops.gelu_bwd(g_in, acts[2 * j - 1], out=ws.get(f"d_pp_pre{j}", g_in.shape, T))
                elif ctx.get("pb_ctx") is not None:
                    d_backbone = ops.mm(g, w[Wn], b_layout=1)                  # gradient w.r.t. the PointBERT output
                    self.pb_trainer.backward(d_backbone.view(Bc, Pn, pb.trans_dim), ctx["pb_ctx"])
                    if self.grad_sync is not None:
                        for nm in self.trainable:
                            if nm.startswith("model.point_backbone."):
                                self._notify(nm)
        if self.grad_sync is not None:                         # everything outside the decoder layers: one last bucket
            self._notify(emb_name)
            self._notify("model.norm.weight")
            for j in range(len(pb.projection_hidden_dim) + 1):
                self._notify(f"model.point_proj.{2 * j}.weight")
                self._notify(f"model.point_proj.{2 * j}.bias")
            self.grad_sync.flush()
        self.ctx = None

    def layer_param_names(self, l):
        p = f"model.layers.{l}."
        return [p + f"self_attn.{n}_proj.weight" for n in "qkvo"] + [p + f"mlp.{n}_proj.weight" for n in ("gate", "up", "down")] + \
               [p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]

    def backward_logits(self, d_logits, hn):
        """lm_head backward: d_hn = d_logits . W ; dW += d_logits^T . hn."""
        W = self.w["lm_head.weight"]
        V, d = W.shape
        Vp = (V + 63) // 64 * 64
        if self.dtype == torch.bfloat16 and d_logits.dim() == 2 and d_logits.stride(0) == Vp and d_logits.stride(1) == 1:
            # padded logits (see logits()): resident [d, V64] transpose of the head, refreshed after every optimizer step
            if self.lm_wT is None or self.lm_wT.shape != (d, Vp):
                self.lm_wT = torch.zeros(d, Vp, dtype=self.dtype, device=self.device)
                self.lm_wT_stale = True
            ver = (W.data_ptr(), self.param_ref["lm_head.weight"]._version if "lm_head.weight" in self.param_ref else W._version)
            if self.lm_wT_stale or self.lm_wT_ver != ver:     # EgoAdamW and load_state_dict flag it; torch optimizers bump the Parameter's _version
                ops.transpose(W, ldo=Vp, out=self.lm_wT)
                self.lm_wT_stale, self.lm_wT_ver = False, ver
            d_hn = ops.mm(torch.as_strided(d_logits, (d_logits.shape[0], Vp), (Vp, 1)), self.lm_wT,
                          out=self.ws.get("d_hn", hn.shape, self.dtype))
        else:
            d_hn = ops.mm(d_logits, W, out=self.ws.get("d_hn", hn.shape, self.dtype), b_layout=1)
        self.reduced_grad.clear()                                # views of the previous step's exchange: consumed by the optimizer since
        self._wgrad("lm_head.weight", d_logits, hn)
        if self.grad_sync is not None:
            self.grad_sync.begin_step()
            self._notify("lm_head.weight")                       # first bucket of the step: travels under the whole backward
            self.grad_sync.flush()
        return d_hn
