#!/usr/bin/env python3
"""transpose A/B: EGOMI_TRANSPOSE_TR=0|1 python tools/debug/transpose_bench.py  (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
for R, C in [(5536, 4096), (5536, 11008), (4096, 4096), (32262 // 8 * 8, 4096)]:
    x = torch.randn(R, C, device="cuda").bfloat16()
    ldo = (R + 63) // 64 * 64
    o = ops.transpose(x, ldo=ldo)
    ok = torch.equal(o[:, :R], x.t()) and (ldo == R or float(o[:, R:].abs().max()) == 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.transpose(x, ldo=ldo, out=o)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"R={R} C={C}: {us:7.1f} us  {2 * R * C * 2 / us / 1e6:5.2f} TB/s  {'ok' if ok else 'MISMATCH'}")
