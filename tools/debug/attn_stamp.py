#!/usr/bin/env python3
"""Where a forward-attention tile-step goes: needs a library built with -DATTN_STAMP (s_memtime stamps in attn_fwd_kernel,
wave 0 of every block; egoscaler_amd/csrc/attention.hip) in place of egoscaler_amd/lib/libegomi.so.  GPU box only."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops, _lib
S, B, H, hd = 692, 8, 32, 128
M, d = B * S, H * hd
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
mask = torch.ones(B, S, device="cuda", dtype=torch.uint8)
L = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(3):
    ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=mask)
torch.cuda.synchronize()
L.egomi_attn_stamp_reset()
N = 10
for _ in range(N):
    ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=mask)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.egomi_attn_stamp_read(buf)
names = ["prologue", "dma issue+vmcnt", "barrier A", "QK+mask", "softmax", "PV", "barrier B", "epilogue"]
tot = sum(buf[i] for i in range(8))
blocks = buf[8]
print(f"blocks stamped {blocks}, ticks per block {tot / blocks:.1f} (s_memtime ticks, 100 MHz: 10 ns each)")
for i, n in enumerate(names):
    print(f"  {n:18s} {buf[i] / blocks:9.1f} ticks/block  {100.0 * buf[i] / tot:5.1f} %")
