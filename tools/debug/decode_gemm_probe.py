#!/usr/bin/env python3
"""Probe: M=256 decode projections on the 256x256 8-phase kernel (EGOMI_GEMM_TILE=8) with every tile K-sliced (rows=1, S)
or in its persistent stream-K form, against the shipped 128x128 split-K path.  GPU box only.
  EGOMI_GEMM_TILE=8 python tools/debug/decode_gemm_probe.py     /     python tools/debug/decode_gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = 256
forced = os.environ.get("EGOMI_GEMM_TILE") == "8"
ws = torch.zeros(160 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = a.float() @ w.float().t()
    line = f"N={N:6d} K={K:6d}:"
    variants = [("auto", dict(workspace=ws))]
    if forced:
        variants += [(f"r1S{S}", dict(workspace=ws, split_k=16 + S)) for S in (2, 3, 4, 5, 6, 8)]
    for name, kw in variants:
        ops.mm(a, w, out=c, **kw)
        err = float((c.float() - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.mm(a, w, out=c, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  {name}:{us:6.1f}us({2*M*N*K/us/1e6:5.0f}TF,{N*K*2/us/1e6:4.2f}TB/s,e{err:.0e})"
    print(line, flush=True)
