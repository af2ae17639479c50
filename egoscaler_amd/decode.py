"""Cached greedy decoding (A13) with static buffers and optional hipGraph capture.

Reference behaviour: model_arch.py:77-108 -> HF generate: one prefill (point encoder + splice,
pointllm.py:112-171) that fills the KV cache, then single-token steps (pointllm.py:255-275).
Here the prefill is Engine.forward_hidden with a kv_sink; each later step runs
  embed -> 32 x {rmsnorm, q/k/v GEMMs, RoPE(pos), kv_append, attn_decode, o_proj, rmsnorm, SwiGLU MLP}
  -> final norm -> lm_head -> argmax
on static buffers.  Every length is a launch argument, so T steps are captured into ONE hipGraph
(BASELINE.json config 5) and replayed with a single launch; token ids never leave the device.
"""
import ctypes
import os

import torch

from . import ops
from ._lib import c_p, c_i, c_f, c_i64, call
from .ops import P, S, dt


def kv_append(k, v, ld, kc, vc, B, Sq, H, hd, Smax, pos0):
    call("egomi_kv_append", P(k), P(v), c_i64(ld), P(kc), P(vc), c_i(B), c_i(Sq), c_i(H), c_i(hd), c_i(Smax), c_i(pos0), c_i(dt(k.dtype)), S())


def attn_decode(q, ld_q, kc, vc, key_mask, out, B, H, hd, Smax, T_len, scale):
    call("egomi_attn_decode", P(q), c_i64(ld_q), P(kc), P(vc), P(key_mask), c_i64(key_mask.stride(0) if key_mask is not None else 0),
         P(out), c_i64(out.stride(0)), c_i(B), c_i(H), c_i(hd), c_i(Smax), c_i(T_len), c_f(scale), c_i(dt(q.dtype)), S())


def argmax_rows(logits, ids, seq=None, pos=0):
    B, V = logits.shape
    call("egomi_argmax_rows", P(logits), c_i64(logits.stride(0)), c_i(B), c_i(V), P(ids), P(seq), c_i64(seq.stride(0) if seq is not None else 0),
         c_i(pos), c_i(dt(logits.dtype)), S())


def sample_rows(logits, scores, seq, pos, rep_from, ids, done, repetition_penalty, temperature, top_k, top_p, do_sample, rng, draw, eos, pad):
    """One step's token choice for the whole batch (include/egomi.h egomi_sample_rows): processed scores (HF's `.scores`) + next token."""
    B, V = logits.shape
    call("egomi_sample_rows", P(logits), c_i64(logits.stride(0)), c_i(B), c_i(V), P(scores), c_i64(scores.stride(0)), P(seq),
         c_i64(seq.stride(0) if seq is not None else 0), c_i(pos), c_i(rep_from), P(ids), P(done), c_f(repetition_penalty), c_f(temperature),
         c_i(int(top_k or 0)), c_f(top_p), c_i(int(bool(do_sample))), P(rng), c_i(draw), c_i64(-1 if eos is None else int(eos)),
         c_i64(0 if pad is None else int(pad)), c_i(dt(logits.dtype)), S())


class Decoder:
    def __init__(self, engine, B, max_len):
        self.eng, self.B, self.Smax = engine, B, max_len
        lm = engine.dims.lm
        L, H, hd, d, Fd, V = lm.num_hidden_layers, lm.num_attention_heads, lm.head_dim, lm.hidden_size, lm.intermediate_size, lm.vocab_size
        T, dev = engine.dtype, engine.device
        self.kc = torch.zeros(L, B, H, max_len, hd, dtype=T, device=dev)
        self.vc = torch.zeros(L, B, H, max_len, hd, dtype=T, device=dev)
        z = lambda *s, dtype=T: torch.zeros(*s, dtype=dtype, device=dev)
        self.x, self.h, self.qkv, self.ao, self.x_mid, self.h2 = z(B, d), z(B, d), z(B, 3 * d), z(B, d), z(B, d), z(B, d)
        self.gu, self.act, self.x_out, self.hn, self.lg = z(B, 2 * Fd), z(B, Fd), z(B, d), z(B, d), z(B, V)
        self.tok = z(B, 1, dtype=torch.int64)
        self.gws = torch.empty(128 << 20, dtype=torch.uint8, device=dev)      # split-K slabs of the skinny decode GEMMs
        # sequences, key mask, generator state and eos flags live in STATIC buffers, so that a captured token loop can be replayed by a later
        # generate() call of the same geometry (run_validation: one capture per (prompt length, new tokens, sampling mode), not one per batch)
        self.seq_buf = z(B, max_len, dtype=torch.int64)
        self.mask_buf = torch.ones(B, max_len, dtype=torch.uint8, device=dev)
        self.rng = z(2, dtype=torch.int64)
        self.done_buf = z(B, dtype=torch.int32)
        self._graphs, self._scores = {}, {}
        self.seq = None
        self.mask = None
        self.pos = 0
        # inference-only resident copies: [Wq;Wk;Wv] and [Wgate;Wup] stacked so one product fills q|k|v (resp. gate|up);
        # fewer, wider launches for the weight-streaming-bound decode step (+0.28 GB per layer of HBM)
        w = engine.w
        if not engine.prepared:
            engine.prepare()
        self.wqkv = [engine.wqkv[l] if l in engine.wqkv else torch.cat([w[f"model.layers.{l}.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
                     for l in range(L)]
        self.wgu = [engine.wgu[l] if l in engine.wgu else engine.stack_gate_up(w[f"model.layers.{l}.mlp.gate_proj.weight"], w[f"model.layers.{l}.mlp.up_proj.weight"])
                    for l in range(L)]                     # interleaved-32 rows when ffn % 32 == 0 (engine.gu_il): gate|up come out interleaved

        # single-token step: the split-K projections leave their fp32 slabs unsummed and the NEXT kernel of the layer sums them
        # while doing its own work (qkv -> RoPE + cache append; o_proj / down_proj -> residual + RMSNorm): three launches and a
        # round trip of each product through HBM fewer per layer.  Slice counts are the library's plan for these shapes (0 =
        # it would not split: that projection keeps the plain path).  EGOMI_DECODE_FUSED=0 switches the whole thing off (A/B).
        self.fused = {"qkv": 0, "o": 0, "down": 0}
        if T == torch.bfloat16 and os.environ.get("EGOMI_DECODE_FUSED", "1") != "0" and B <= 512:
            self.fused["qkv"] = ops.mm_slabs(self.h, self.wqkv[0], self.qkv, self.gws, count_only=True)
            self.fused["o"] = ops.mm_slabs(self.ao, w["model.layers.0.self_attn.o_proj.weight"], self.x_mid, self.gws, count_only=True)
            self.fused["down"] = ops.mm_slabs(self.act, w["model.layers.0.mlp.down_proj.weight"], self.x, self.gws, count_only=True)

    # -- prefill -------------------------------------------------------------------------------------
    def _set_inputs(self, input_ids, mask, total_new):
        """Prompt ids and key mask into the static buffers (`seq` is a [B, S0 + total_new] view of rows that are Smax long)."""
        B, S0 = input_ids.shape
        if S0 + total_new > self.Smax:
            raise ValueError("prompt + new tokens exceed the decoder's cache length")
        self.mask_buf.fill_(1)
        self.mask_buf[:, :S0] = mask.to(torch.uint8)
        self.mask = self.mask_buf
        self.seq_buf.zero_()
        self.seq_buf[:, :S0] = input_ids
        self.seq = self.seq_buf[:, :S0 + total_new]

    def _sink(self, l, qkv, B, Sq):
        lm = self.eng.dims.lm
        H, hd, d = lm.num_attention_heads, lm.head_dim, lm.hidden_size
        kv_append(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv.stride(0), self.kc[l], self.vc[l], B, Sq, H, hd, self.Smax, 0)

    def prefill(self, input_ids, attention_mask, point_clouds, fps_start, total_new):
        B, S0 = input_ids.shape
        dev = self.eng.device
        mask = torch.ones(B, S0, dtype=torch.bool, device=dev) if attention_mask is None else attention_mask.to(dev).bool()
        self._set_inputs(input_ids, mask, total_new)
        hn = self.eng.forward_hidden(input_ids, mask, point_clouds, fps_start, save=False, kv_sink=self._sink)
        self.pos = S0
        last = hn.view(B, S0, -1)[:, -1].contiguous()
        ops.mm(last, self.eng.w["lm_head.weight"], out=self.lg)
        return self.lg

    def prefill_chunked(self, input_ids, attention_mask, point_clouds, fps_start, total_new, chunk=16):
        """prefill() for large batches (config 5: bs=256): the prompt pass runs `chunk` samples at a time (its activations are
        what limits the batch, not the cache) and every chunk appends its K/V rows to its own slice of the static cache."""
        B, S0 = input_ids.shape
        eng, dev = self.eng, self.eng.device
        lm = eng.dims.lm
        H, hd, d = lm.num_attention_heads, lm.head_dim, lm.hidden_size
        mask = torch.ones(B, S0, dtype=torch.bool, device=dev) if attention_mask is None else attention_mask.to(dev).bool()
        self._set_inputs(input_ids, mask, total_new)
        for b0 in range(0, B, chunk):
            b1 = min(B, b0 + chunk)

            def sink(l, qkv, Bc, Sq, b0=b0):
                for i in range(Bc):               # a batch slice of the [L,B,H,Smax,hd] cache is not contiguous over samples: one append each
                    kv_append(qkv[i * Sq:(i + 1) * Sq, d:2 * d], qkv[i * Sq:(i + 1) * Sq, 2 * d:], qkv.stride(0), self.kc[l, b0 + i], self.vc[l, b0 + i],
                              1, Sq, H, hd, self.Smax, 0)
            pcs = None if point_clouds is None else point_clouds[b0:b1]
            st = None if fps_start is None else fps_start[b0:b1]
            hn = eng.forward_hidden(input_ids[b0:b1], mask[b0:b1], pcs, st, save=False, kv_sink=sink)
            last = hn.view(b1 - b0, S0, -1)[:, -1].contiguous()
            ops.mm(last, eng.w["lm_head.weight"], out=self.lg[b0:b1])
        self.pos = S0
        return self.lg

    # -- one decode step on static buffers: consumes self.tok, leaves logits in self.lg ---------------------
    def step(self, pos):
        eng = self.eng
        w, lm = eng.w, eng.dims.lm
        B, d, Fd, H, hd, L = self.B, lm.hidden_size, lm.intermediate_size, lm.num_attention_heads, lm.head_dim, lm.num_hidden_layers
        ops.embed_splice(self.tok, w["model.embed_tokens.weight"], None, None, eng.dims.pb.point_token_len, out=self.x.view(B, 1, d))
        x = self.x
        scale = hd ** -0.5
        fq, fo, fd = self.fused["qkv"], self.fused["o"], self.fused["down"]
        normed = False                                      # self.h already holds this layer's input norm (written by the previous layer's tail)
        for l in range(L):
            p = f"model.layers.{l}."
            if not normed:
                ops.rmsnorm(x, w[p + "input_layernorm.weight"], lm.rms_norm_eps, out=self.h)
            if fq:
                n = ops.mm_slabs(self.h, self.wqkv[l], self.qkv, self.gws)
                ops.qkv_finish(self.gws, n, self.qkv, eng.cos, eng.sin, pos, self.kc[l], self.vc[l], B, H, hd, self.Smax)
            else:
                ops.mm(self.h, self.wqkv[l], out=self.qkv, workspace=self.gws)
                ops.rope_(self.qkv, eng.cos, eng.sin, B, 1, pos, 2 * H, hd, 3 * d)
                kv_append(self.qkv[:, d:2 * d], self.qkv[:, 2 * d:], 3 * d, self.kc[l], self.vc[l], B, 1, H, hd, self.Smax, pos)
            attn_decode(self.qkv, 3 * d, self.kc[l], self.vc[l], self.mask, self.ao, B, H, hd, self.Smax, pos + 1, scale)
            if fo:
                n = ops.mm_slabs(self.ao, w[p + "self_attn.o_proj.weight"], self.x_mid, self.gws)
                ops.slabs_rmsnorm(self.gws, n, x, w[p + "post_attention_layernorm.weight"], lm.rms_norm_eps, self.x_mid, self.h2)
            else:
                ops.mm(self.ao, w[p + "self_attn.o_proj.weight"], out=self.x_mid, residual=x, workspace=self.gws)
                ops.rmsnorm(self.x_mid, w[p + "post_attention_layernorm.weight"], lm.rms_norm_eps, out=self.h2)
            ops.mm(self.h2, self.wgu[l], out=self.gu, workspace=self.gws)
            if eng.gu_il:
                ops.swiglu_il(self.gu, self.act)
            else:
                ops.swiglu(self.gu[:, :Fd], self.gu[:, Fd:], self.act)
            if fd:                                          # x is not an input of this product: the tail writes the new residual stream into it
                n = ops.mm_slabs(self.act, w[p + "mlp.down_proj.weight"], x, self.gws)
                last = l + 1 == L
                ops.slabs_rmsnorm(self.gws, n, self.x_mid, w["model.norm.weight"] if last else w[f"model.layers.{l + 1}.input_layernorm.weight"],
                                  lm.rms_norm_eps, x, self.hn if last else self.h)
                normed = True
            else:
                ops.mm(self.act, w[p + "mlp.down_proj.weight"], out=x, residual=self.x_mid, workspace=self.gws)
                normed = False
        if not normed:
            ops.rmsnorm(x, w["model.norm.weight"], lm.rms_norm_eps, out=self.hn)
        ops.mm(self.hn, w["lm_head.weight"], out=self.lg)

    def sample(self, T_new, do_sample=True, temperature=1.0, top_k=50, top_p=0.95, repetition_penalty=1.0, eos=None, pad=None, seed=None,
               use_graph=True):
        """After prefill(): T_new steps of HF generate's token loop (model_arch.py:82-108 -> GenerationMixin: logits processors, warpers,
        multinomial / arg-max, eos bookkeeping), every step one egomi_sample_rows launch + one cached decode step, all of them captured
        into ONE hipGraph (the draw counter and `pos` are launch constants, seed and done flags live in device memory).
        Returns (sequences [B, S0+T_new], processed scores fp32 [T_new, B, V]); rows that emitted `eos` continue with `pad`."""
        S0, dev = self.pos, self.eng.device
        V = self.lg.shape[1]
        sc_buf = self._scores.get(T_new)
        if sc_buf is None:
            sc_buf = self._scores[T_new] = torch.empty(T_new, self.B, V, dtype=torch.float32, device=dev)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # the CPU default generator: torch.manual_seed() makes a run repeatable
        self.rng.copy_(torch.tensor([int(seed), 0], dtype=torch.int64))
        self.done_buf.zero_()
        self.done = self.done_buf if eos is not None else None
        kw = dict(repetition_penalty=float(repetition_penalty or 1.0), temperature=float(temperature or 1.0), top_k=int(top_k or 0),
                  top_p=float(1.0 if top_p is None else top_p), do_sample=bool(do_sample), rng=self.rng, eos=eos, pad=pad)

        def steps():
            for t in range(T_new):
                sample_rows(self.lg, sc_buf[t], self.seq, S0 + t, 0, self.tok.view(-1), self.done, draw=t, **kw)
                if t + 1 < T_new:
                    self.step(S0 + t)
        if not use_graph:
            steps()
        else:
            # the captured loop depends on the prompt length, the number of steps and the sampling parameters only (every buffer it touches is
            # static, the seed and the eos flags are device memory): a later call with the same key replays it
            key = (S0, T_new, kw["repetition_penalty"], kw["temperature"], kw["top_k"], kw["top_p"], kw["do_sample"], eos, pad)
            g = self._graphs.get(key)
            if g is None:
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        steps()
                torch.cuda.current_stream().wait_stream(side)
                self._graphs[key] = g
            g.replay()
            self.graph = g
        self.pos = S0 + T_new
        return self.seq, sc_buf

    def greedy(self, T_new, use_graph=True, keep_scores=True):
        """After prefill(): T_new greedy tokens.  Returns (sequences [B,S0+T], scores list or None)."""
        S0 = self.pos
        scores = [] if keep_scores else None
        if not use_graph:
            for t in range(T_new):
                if keep_scores:
                    scores.append(self.lg.float().clone())
                argmax_rows(self.lg, self.tok.view(-1), self.seq, S0 + t)
                if t + 1 < T_new:
                    self.step(S0 + t)
            self.pos = S0 + T_new
            return self.seq, scores
        sc_buf = torch.zeros(T_new, self.B, self.lg.shape[1], dtype=torch.float32, device=self.eng.device) if keep_scores else None
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for t in range(T_new):
                    if keep_scores:
                        ops.cast(self.lg, torch.float32, out=sc_buf[t])
                    argmax_rows(self.lg, self.tok.view(-1), self.seq, S0 + t)
                    if t + 1 < T_new:
                        self.step(S0 + t)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        self.graph = g
        self.pos = S0 + T_new
        return self.seq, ([sc_buf[t] for t in range(T_new)] if keep_scores else None)
