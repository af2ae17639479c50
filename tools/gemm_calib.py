#!/usr/bin/env python3
"""Calibration only (not a product dependency): what the vendor GEMM (torch.mm -> hipBLASLt) reaches on the bench
step's shapes, next to egomi_gemm, so the headroom of the hand-written kernel is known.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5536
shapes = [(4096, 4096), (12288, 4096), (11008, 4096), (4096, 11008), (4096, 12288), (8192, 8192)]
for N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = {}
    for name, fn in (("egomi", lambda: ops.mm(a, w, out=c)), ("vendor", lambda: torch.mm(a, w.t(), out=c))):
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        med = sorted(ts)[len(ts) // 2]
        out[name] = 2 * M * N * K / med / 1e9
    print(f"M={M} N={N:6d} K={K:6d}  egomi {out['egomi']:7.1f}  vendor {out['vendor']:7.1f} TFLOP/s")
