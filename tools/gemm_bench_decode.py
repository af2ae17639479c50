#!/usr/bin/env python3
"""Skinny-M (decode) GEMM micro-benchmark: M rows (default 256; 8 = the reference's evaluation batch) against LLaMA-7B weights, split-K sweep,
rotating weights (each launch reads a matrix that is not cache-resident).  python tools/gemm_bench_decode.py [M]   GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32262, 4096)]:          # qkv, o, gate|up, down, lm_head
    a = torch.randn(M, K, device="cuda").bfloat16()
    nw = max(2, -(-(600 << 20) // (N * K * 2)))
    wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
    w = wl[0]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = ops.mm(a, w)
    line = f"N={N:6d} K={K:6d}:"
    for sk in (1, 2, 3, 4, 6, 8, 12, 16, 0):
        kw = {} if sk == 1 else dict(workspace=ws, split_k=sk)
        ops.mm(a, w, out=c, **kw)
        err = float((c.float() - ref.float()).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            ops.mm(a, wl[i % nw], out=c, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  sk{sk}:{us:6.1f}us({N*K*2/us/1e6:4.2f}TB/s,e{err:.0e})"
    print(line)

