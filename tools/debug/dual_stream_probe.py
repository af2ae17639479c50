#!/usr/bin/env python3
"""Would two half-batches on two streams fill the ragged last rounds of the big GEMMs?  One decoder layer's eight products (forward + data
gradients, frozen-LLM shapes) for M = 5536 on one stream against the same products for two halves of M = 2768 on two streams, rotating
weights (cold, as in the step).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops

shapes = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (11008, 4096), (4096, 22016), (4096, 4096), (4096, 12288)]   # (N, K) in layer order fwd then bwd
L = 6                                                   # distinct "layers" of weights: > 256 MB per shape set
W = [[(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for (N, K) in shapes] for _ in range(L)]
ws = [torch.zeros(256 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]


def run(M, streams):
    A = {K: [(torch.randn(M, K, device="cuda")).bfloat16() for _ in streams] for K in (4096, 11008, 12288, 22016)}
    C = {N: [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in streams] for N in (4096, 11008, 12288, 22016)}
    def once():
        for l in range(L):
            for i, (N, K) in enumerate(shapes):
                for si, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        ops.mm(A[K][si], W[l][i], out=C[N][si], workspace=ws[si])
    once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        e0.record()
        for st in streams:
            st.wait_event(e0)
        once()
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / L)
    return sorted(ts)[len(ts) // 2]


s0 = torch.cuda.current_stream()
for rnd in range(3):
    t1 = run(5536, [s0])
    t2 = run(2768, [torch.cuda.Stream(), torch.cuda.Stream()])
    t3 = run(2768, [s0])
    print(f"one stream M=5536: {t1*1e3:8.1f} us/layer   two streams 2 x M=2768: {t2*1e3:8.1f} us/layer   one stream M=2768 (x2 = {2*t3*1e3:8.1f}): {t3*1e3:8.1f}", flush=True)
