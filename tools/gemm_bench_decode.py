#!/usr/bin/env python3
"""Skinny-M (decode) GEMM micro-benchmark: M=256 against LLaMA-7B weights, split-K sweep.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops
M = 256
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32262, 4096)]:          # qkv, o, gate|up, down, lm_head
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = ops.mm(a, w)
    line = f"N={N:6d} K={K:6d}:"
    for sk in (1, 2, 3, 4, 6, 8, 12, 16, 0):
        kw = {} if sk == 1 else dict(workspace=ws, split_k=sk)
        ops.mm(a, w, out=c, **kw)
        err = float((c.float() - ref.float()).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.mm(a, w, out=c, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  sk{sk}:{us:6.1f}us({N*K*2/us/1e6:4.2f}TB/s,e{err:.0e})"
    print(line)

