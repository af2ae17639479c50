#!/usr/bin/env python3
"""Micro-benchmark of egomi_gemm on the LLaMA-7B shapes of the bench step (random bf16 operands,
interleaved rounds in one process, HIP events on the launch stream).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5536
shapes = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (1024, 384), (4096, 2048)]
bufs = {}
for N, K in shapes:
    a = (torch.randn(M, K, device="cuda") * 1.0).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bufs[(N, K)] = (a, w, c)
res = {s: [] for s in shapes}
for rnd in range(6):
    for s in shapes:
        a, w, c = bufs[s]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.mm(a, w, out=c)
        e0.record()
        for _ in range(5):
            ops.mm(a, w, out=c)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[s].append(e0.elapsed_time(e1) / 5)
for (N, K), t in res.items():
    t = sorted(t)
    med = t[len(t) // 2]
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M} N={N:6d} K={K:6d} tiles={tiles:5d} median {med*1e3:8.1f} us  {2*M*N*K/med/1e9:8.1f} TFLOP/s  (min {2*M*N*K/t[0]/1e9:.1f})")
