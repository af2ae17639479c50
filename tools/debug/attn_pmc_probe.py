"""A few launches of the three attention kernels at the bench step's shape, for rocprofv3 --pmc passes (tools/debug/pmc_sum_attn.py sums them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
B, S, H, hd = 8, 692, 32, 128
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
    ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=True, key_mask=mask)
torch.cuda.synchronize()
