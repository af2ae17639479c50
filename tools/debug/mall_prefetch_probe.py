#!/usr/bin/env python3
"""Do the M = 256 decode projections run faster when their weights were read (by anything) just before — i.e. served by the 256-MB Infinity Cache
instead of HBM?  (a) rotating weights (cold), (b) the same matrix every launch (hot), (c) rotating, each launch preceded by a read-only pass over
the matrix it is about to multiply (torch.sum of an int32 view), GEMM timed alone by events.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M = 256
for N, K in [(22016, 4096), (12288, 4096), (4096, 4096), (32262, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    nw = max(3, -(-(800 << 20) // (N * K * 2)))
    wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(128 << 20, dtype=torch.uint8, device="cuda")
    def timed(fn_pre, pick):
        ts = []
        for i in range(12):
            w = pick(i)
            if fn_pre is not None:
                fn_pre(w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.mm(a, w, out=c, workspace=ws); e1.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); return ts[len(ts) // 2]
    cold = timed(None, lambda i: wl[i % nw])
    hot = timed(None, lambda i: wl[0])
    touch = timed(lambda w: w.view(torch.int32).sum(), lambda i: wl[i % nw])
    print(f"M=256 N={N:6d} K={K}: cold {cold:6.1f} us   same matrix {hot:6.1f} us   read-just-before {touch:6.1f} us   ({N*K*2/1e6:.0f} MB of weights)", flush=True)
