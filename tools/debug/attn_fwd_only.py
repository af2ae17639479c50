#!/usr/bin/env python3
"""attn_fwd / attn_bwd alone at the bench shape, a few launches (for rocprofv3 --pmc passes).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
S, B, H, hd = 692, 8, 32, 128
M, d = B * S, H * hd
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
dout = (torch.randn(M, d, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
delta = torch.empty_like(lse)
mask = torch.ones(B, S, device="cuda", dtype=torch.uint8)
for _ in range(4):
    ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=mask)
    if "bwd" in sys.argv:
        ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=True, key_mask=mask)
torch.cuda.synchronize()
