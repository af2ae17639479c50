"""Oracle for SURVEY.md §8a rows A11-A13: the LLaMA decoder stack, lm_head, the EgoScaler loss and
greedy decoding.  Test infrastructure only (see oracle/__init__.py).

The reference reaches this arithmetic through HuggingFace `transformers` (third-party, NOT under
/root/reference; pinned by the reference at git cae78c46 ~ 4.28.0.dev, pyproject.toml:21, but its
own code needs a much newer API — SURVEY.md §8c).  The algorithm restated here is the published
LLaMA forward as implemented by the installed transformers 5.15.0
(models/llama/modeling_llama.py:53-71 RMSNorm, :112-127 rotary cos/sin, :130-160 rotate_half
application, :191-214 eager attention, :174-176 SwiGLU, :284-325 decoder layer, :413 final norm),
anchored on the reference's call sites pointllm/model/pointllm.py:173-178,227-228 and
train.py:174-181.  Golden vectors produced by those classes pin it (tests/golden).
"""
import math

import torch
import torch.nn.functional as F


def rms_norm(x, w, eps):
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_cos_sin(S, head_dim, theta, offset=0, dtype=torch.float32):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    pos = torch.arange(offset, offset + S, dtype=torch.float32)
    freqs = pos[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """q,k [B,H,S,hd]; cos/sin [S,hd]."""
    return q * cos + _rot_half(q) * sin, k * cos + _rot_half(k) * sin


def causal_bias(attention_mask, S, dtype, past=0):
    """Additive mask [B,1,S,past+S]: causal + key padding (what create_causal_mask yields for eager).
    attention_mask: None or [B, past+S] bool/int (1 = keep)."""
    T = past + S
    q = torch.arange(past, T)[:, None]
    k = torch.arange(T)[None, :]
    keep = (k <= q)[None, None]
    if attention_mask is not None:
        keep = keep & attention_mask.to(torch.bool)[:, None, None, :]
    return torch.where(keep, torch.zeros((), dtype=dtype), torch.full((), torch.finfo(dtype).min, dtype=dtype))


def attention(sd, p, x, cos, sin, bias, H, kv=None):
    B, S, d = x.shape
    hd = d // H
    q = F.linear(x, sd[p + "q_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
    v = F.linear(x, sd[p + "v_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if kv is not None:
        if kv.get("k") is not None:
            k = torch.cat([kv["k"], k], dim=2)
            v = torch.cat([kv["v"], v], dim=2)
        kv["k"], kv["v"] = k, v
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    w = w + bias
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, d)
    return F.linear(o, sd[p + "o_proj.weight"])


def mlp(sd, p, x):
    return F.linear(F.silu(F.linear(x, sd[p + "gate_proj.weight"])) * F.linear(x, sd[p + "up_proj.weight"]),
                    sd[p + "down_proj.weight"])


def decoder_stack(sd, x, attention_mask, lm, kv_cache=None, taps=None):
    """inputs_embeds [B,S,d] -> final-normed hidden [B,S,d].  kv_cache: None or list of dicts."""
    B, S, d = x.shape
    past = 0
    if kv_cache is not None and kv_cache[0].get("k") is not None:
        past = kv_cache[0]["k"].shape[2]
    cos, sin = rope_cos_sin(S, lm.head_dim, lm.rope_theta, offset=past, dtype=x.dtype)
    bias = causal_bias(attention_mask, S, x.dtype, past)
    for i in range(lm.num_hidden_layers):
        p = f"model.layers.{i}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"], lm.rms_norm_eps)
        x = x + attention(sd, p + "self_attn.", h, cos, sin, bias, lm.num_attention_heads,
                          None if kv_cache is None else kv_cache[i])
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], lm.rms_norm_eps)
        x = x + mlp(sd, p + "mlp.", h)
        if taps is not None:
            taps[f"layer{i}"] = x
    return rms_norm(x, sd["model.norm.weight"], lm.rms_norm_eps)


def traj_loss(logits, tokens, prompt_len, pad_id):
    """EgoScaler's loss, train.py:174-181: logits[:, Lp-1:-1] vs tokens[:, Lp:], CE ignoring pad."""
    lg = logits[:, prompt_len - 1:-1, :]
    tg = tokens[:, prompt_len:]
    return F.cross_entropy(lg.reshape(-1, lg.shape[-1]), tg.flatten(), ignore_index=pad_id)
