#!/usr/bin/env python3
"""egomi_gemm timing for explicit M,N,K triples: python tools/gemm_bench_shapes.py M,N,K [M,N,K ...]  (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

for arg in sys.argv[1:]:
    M, N, K = (int(x) for x in arg.split(","))
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ts = []
    for rnd in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.mm(a, w, out=c)
        e0.record()
        for _ in range(5):
            ops.mm(a, w, out=c)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    med = sorted(ts[1:])[len(ts[1:]) // 2]
    print(f"M={M:5d} N={N:6d} K={K:6d} median {med*1e3:8.1f} us  {2*M*N*K/med/1e9:8.1f} TFLOP/s")
