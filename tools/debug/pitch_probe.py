#!/usr/bin/env python3
"""Does the row pitch of the operands matter for the M = 256 decode projections?  x [256, K] and W [N, K] with K = 4096 have 8-KB rows: every block reads the
same K-slice of x at the same time, and a K-slice of either operand is a set of 128-B pieces exactly 8 KB apart (channel / bank aliasing?).  Times the same
product with the operands stored at pitch K and at pitch K + pad elements (views of wider buffers).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = 256
ws = torch.zeros(128 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(22016, 4096), (12288, 4096), (32262, 4096), (4096, 11008)]:
    nw = max(3, -(-(800 << 20) // (N * K * 2)))
    for pad_a, pad_w in [(0, 0), (64, 0), (0, 64), (64, 64), (192, 192)]:
        a = torch.randn(M, K + pad_a, device="cuda").bfloat16()[:, :K]
        wl = [(torch.randn(N, K + pad_w, device="cuda") * 0.02).bfloat16()[:, :K] for _ in range(nw)]
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.mm(a, wl[0], out=c, workspace=ws)
            e0.record()
            for i in range(8):
                ops.mm(a, wl[(rnd * 8 + i + 1) % nw], out=c, workspace=ws)
            e1.record(); torch.cuda.synchronize()
            if rnd: ts.append(e0.elapsed_time(e1) / 8 * 1e3)
        ts.sort()
        print(f"M=256 N={N:6d} K={K:6d} pitch x +{pad_a:3d} W +{pad_w:3d}: {ts[len(ts)//2]:6.1f} us", flush=True)
        del wl
