#!/usr/bin/env python3
"""Probe for `rocprofv3 --pmc`: persistent vs per-tile 8-phase GEMM with cold (rotating) weights.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = 5536
for N, K in [(4096, 4096), (4096, 2048), (12288, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(max(2, (640 << 20) // (N * K * 2)))]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for pers in (True, False):
        for i in range(12):
            ops.mm(a, ws[i % len(ws)], out=c, persistent=pers)
    torch.cuda.synchronize()
    del ws
print("done")
