#!/usr/bin/env python3
"""A few cold-weight launches of the N = 4096 products at M = 5536 in the library's default forms, for rocprofv3 --pmc passes (tools/debug/pmc_sum_tall.py sums them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = 5536
ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(4096, 12288), (4096, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(4)]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for i in range(8):
        ops.mm(a, wl[i % 4], out=c, workspace=ws)
torch.cuda.synchronize()
