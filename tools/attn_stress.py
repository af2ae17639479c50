#!/usr/bin/env python3
"""Randomised screen of the fused attention kernels (forward hd 128 / 64, backward hd 128 with and without the fused inverse
RoPE) against an fp32 torch evaluation: random B, S (1..1500), H, causal / key-padding masks, repeated launches.  GPU box only."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
bad = 0


def ref_attn(qkv, B, S, H, hd, causal, km, dout=None):
    x = qkv.float().view(B, S, 3, H, hd).clone().requires_grad_(dout is not None)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    sc = (q @ k.transpose(-1, -2)) * hd ** -0.5
    keep = torch.ones(S, S, dtype=torch.bool, device=qkv.device)
    if causal:
        keep = torch.tril(keep)
    keep = keep[None, None].expand(B, 1, S, S)
    if km is not None:
        keep = keep & km.bool()[:, None, None, :]
    sc = sc.masked_fill(~keep, float("-inf"))
    p = torch.softmax(sc, -1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ v).transpose(1, 2).reshape(B * S, H * hd)
    if dout is None:
        return o, None
    (g,) = torch.autograd.grad(o, x, dout.float())
    return o.detach(), g.reshape(B * S, 3 * H * hd)


for it in range(n):
    hd = rng.choice([128, 128, 64])
    B, H = rng.randint(1, 3), rng.randint(1, 4)
    S = rng.choice([rng.randint(1, 70), rng.randint(70, 300), rng.randint(300, 1500), 692, 513])
    causal = rng.random() < 0.7 if hd == 128 else rng.random() < 0.3
    masked = rng.random() < 0.5
    d = H * hd
    qkv = (torch.randn(B * S, 3 * d, device="cuda") * 0.7).bfloat16()
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
        cut = rng.randint(0, max(0, S - 1))
        km[-1, S - cut:] = 0                                   # right padding on the last sample (at least one visible key)
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    dout = (torch.randn(B * S, d, device="cuda") * 0.1).bfloat16()
    o_ref, g_ref = ref_attn(qkv, B, S, H, hd, causal, km, dout if hd == 128 else None)
    worst = 0.0
    for rep in range(2):
        ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=km)
        worst = max(worst, float((out.float() - o_ref).abs().max()) / (float(o_ref.abs().max()) + 1e-6))
        if hd == 128:
            dqkv = torch.zeros_like(qkv)
            delta = torch.empty_like(lse)
            ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=causal, key_mask=km)
            worst = max(worst, float((dqkv.float() - g_ref).abs().max()) / (float(g_ref.abs().max()) + 1e-6))
            cos, sin = ops.rope_tables(S + 3, hd, 10000.0)
            cos, sin = cos.cuda(), sin.cuda()
            plain = dqkv.clone()
            ops.rope_(plain, cos, sin, B * S, S, 0, 2 * H, hd, 3 * d, inverse=True)
            fused = torch.zeros_like(qkv)
            ops.attn_bwd(qkv, out, lse, dout, fused, delta, B, S, H, hd, hd ** -0.5, causal=causal, key_mask=km, rope=(cos, sin))
            if not torch.equal(fused, plain):
                worst = 1.0
    ok = worst < 3e-2
    bad += (not ok)
    if not ok or it % 8 == 0:
        print(f"[{it}] hd={hd} B={B} H={H} S={S} causal={causal} masked={masked}: worst {worst:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("FAILED" if bad else f"all {n} cases ok")
sys.exit(1 if bad else 0)
