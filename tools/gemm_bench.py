#!/usr/bin/env python3
"""Micro-benchmark of egomi_gemm on the LLaMA-7B shapes of the bench step (random bf16 operands, interleaved rounds in one
process, HIP events on the launch stream).  GPU box only.   python tools/gemm_bench.py [M] [cold]

`cold`: every launch multiplies a DIFFERENT weight matrix out of a rotating set larger than the 256-MB Infinity Cache, as the
training step does (32 layers, each weight read once per pass).  Without it the same operands are re-used launch after
launch and stay cache-resident: round 2 measured the persistent kernel +1...5 % that way and -5 % in the real step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5536
COLD = "cold" in sys.argv[1:]
shapes = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (4096, 2048)]
bufs = {}
for N, K in shapes:
    a = (torch.randn(M, K, device="cuda") * 1.0).bfloat16()
    nw = max(1, -(-(640 << 20) // (N * K * 2))) if COLD else 1          # >= 640 MB of distinct weights per shape
    w = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bufs[(N, K)] = (a, w, c)
# A/B in one process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24): persistent form vs non-persistent + combine launch
res = {(s, p): [] for s in shapes for p in (True, False)}
for rnd in range(7):
    for s in shapes:
        for pers in (True, False):
            a, w, c = bufs[s]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.mm(a, w[0], out=c, persistent=pers)
            e0.record()
            for i in range(5):
                ops.mm(a, w[(rnd * 5 + i + 1) % len(w)], out=c, persistent=pers)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[(s, pers)].append(e0.elapsed_time(e1) / 5)
for (N, K) in shapes:
    line = f"M={M} N={N:6d} K={K:6d} tiles256={((M + 255) // 256) * ((N + 255) // 256):5d}"
    for pers in (True, False):
        t = sorted(res[((N, K), pers)])
        med = t[len(t) // 2]
        line += f" | {'persistent' if pers else 'per-tile  '} median {med*1e3:7.1f} us {2*M*N*K/med/1e9:7.1f} TF (best {2*M*N*K/t[0]/1e9:.1f})"
    print(line)
