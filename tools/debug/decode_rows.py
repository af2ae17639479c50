#!/usr/bin/env python3
"""Debug: which op of the decode step makes rows that share a prompt diverge?  (GPU box only)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import synth, ops
from egoscaler_amd.config import dims_7b
from egoscaler_amd.decode import Decoder, kv_append, attn_decode
from egoscaler_amd.pointllm import TrajPointLLMForCausalLM

B, T = 256, 4
dims = dims_7b(); dims.lm.num_hidden_layers = 2
dev = torch.device("cuda")
args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=256, model_name=None)
m = TrajPointLLMForCausalLM(args, dims, None, device=dev, dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(7)
with torch.no_grad():
    for n, p in list(m.named_parameters()) + list(m.named_buffers()):
        leaf = n.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked": continue
        if leaf == "running_var" or (leaf == "weight" and p.dim() == 1): p.fill_(1.0)
        elif leaf == "running_mean": p.zero_()
        else:
            fan = p[0].numel() if p.dim() > 1 else p.numel()
            p.copy_(torch.empty(p.shape, dtype=torch.float32, device=dev).normal_(0, 0.02 if fan >= 1024 else min(0.35, fan ** -0.5), generator=g))
m.engine.prepared = False
m.eval()
eng = m.engine
toks, masks, Lp = synth.synth_batch(dims, 4, text_len=16, num_steps=20, max_traj_token=160)
ids = toks[:, :Lp].repeat(B // 4, 1).to(dev)
pcs = torch.stack([synth.synth_cloud(dims, i) for i in range(4)]).repeat(B // 4, 1, 1).to(dev)
dec = Decoder(eng, B, Lp + T)
dec.prefill_chunked(ids, None, pcs, torch.zeros(B, dtype=torch.int32, device=dev), T, chunk=16)
def rows_same(name, t):
    t2 = t.view(B, -1)
    ref = t2[:4].repeat(B // 4, 1)
    bad = (t2 != ref).any(1).nonzero().flatten()
    print(f"{name:10s} rows differing from their prompt's first row: {bad.numel()}", bad[:12].tolist(), flush=True)
rows_same("prefill lg", dec.lg)
rows_same("kc[0]", dec.kc[0].reshape(B, -1))
rows_same("kc[1]", dec.kc[1].reshape(B, -1))
from egoscaler_amd.decode import argmax_rows
argmax_rows(dec.lg, dec.tok.view(-1), dec.seq, Lp)
rows_same("tok", dec.tok)
w, lm = eng.w, dims.lm
d, Fd, H, hd = lm.hidden_size, lm.intermediate_size, lm.num_attention_heads, lm.head_dim
pos = Lp
ops.embed_splice(dec.tok, w["model.embed_tokens.weight"], None, None, dims.pb.point_token_len, out=dec.x.view(B, 1, d))
x = dec.x
rows_same("embed", x)
for l in range(2):
    p = f"model.layers.{l}."
    ops.rmsnorm(x, w[p + "input_layernorm.weight"], lm.rms_norm_eps, out=dec.h); rows_same("h", dec.h)
    ops.mm(dec.h, dec.wqkv[l], out=dec.qkv, workspace=dec.gws); rows_same("qkv", dec.qkv)
    ops.rope_(dec.qkv, eng.cos, eng.sin, B, 1, pos, 2 * H, hd, 3 * d); rows_same("rope", dec.qkv)
    kv_append(dec.qkv[:, d:2 * d], dec.qkv[:, 2 * d:], 3 * d, dec.kc[l], dec.vc[l], B, 1, H, hd, dec.Smax, pos)
    rows_same("kc+", dec.kc[l].reshape(B, -1))
    attn_decode(dec.qkv, 3 * d, dec.kc[l], dec.vc[l], dec.mask, dec.ao, B, H, hd, dec.Smax, pos + 1, hd ** -0.5); rows_same("ao", dec.ao)
    ops.mm(dec.ao, w[p + "self_attn.o_proj.weight"], out=dec.x_mid, residual=x, workspace=dec.gws); rows_same("x_mid", dec.x_mid)
    ops.rmsnorm(dec.x_mid, w[p + "post_attention_layernorm.weight"], lm.rms_norm_eps, out=dec.h2); rows_same("h2", dec.h2)
    ops.mm(dec.h2, dec.wgu[l], out=dec.gu, workspace=dec.gws); rows_same("gu", dec.gu)
    ops.swiglu(dec.gu[:, :Fd], dec.gu[:, Fd:], dec.act); rows_same("act", dec.act)
    ops.mm(dec.act, w[p + "mlp.down_proj.weight"], out=x, residual=dec.x_mid, workspace=dec.gws); rows_same("x", x)
ops.rmsnorm(x, w["model.norm.weight"], lm.rms_norm_eps, out=dec.hn); rows_same("hn", dec.hn)
ops.mm(dec.hn, w["lm_head.weight"], out=dec.lg); rows_same("lg", dec.lg)
