#!/usr/bin/env python3
"""Sweep explicit tail plans (rows of M-tiles cut into S K-slices) of the per-tile 8-phase GEMM against the library's own
plan, cold weights.  GPU box only.  python tools/debug/tail_plan_sweep.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5536
ws = torch.zeros((160 << 20) // 4, dtype=torch.float32, device="cuda")
for N, K in [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (12288, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(max(2, (640 << 20) // (N * K * 2)))]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def t(**kw):
        ops.mm(a, w[0], out=c, **kw)
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(6):
                ops.mm(a, w[(r * 6 + i) % len(w)], out=c, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 6 * 1e3)
        return sorted(ts)[1]
    base = t(persistent=False)
    res = []
    tm = (M + 255) // 256
    for rows in range(1, min(tm, 10) + 1):
        for S in (2, 3, 4, 5, 6, 8):
            if (K // 64) // S < 8:
                continue
            res.append((t(workspace=ws, split_k=rows * 16 + S), rows, S))
    res.sort()
    print(f"M={M} N={N} K={K}: library plan {base:7.1f} us | no tail {t(workspace=None) if False else 0:.0f} | best explicit: " +
          ", ".join(f"r{r}S{s} {u:6.1f}" for u, r, s in res[:5]), flush=True)
