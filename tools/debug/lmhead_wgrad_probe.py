#!/usr/bin/env python3
"""lm_head wgrad of the bench step: dW [V=32262, d=4096] (fp32) = dlogits^T [V, 1280] . hn^T [4096, 1280]^T, K = 1280 span rows (as the
engine runs it: both operands transposed into K-contiguous buffers).  128x128 two-stage kernel (the library's choice: K < 2048)
vs the 256x256 kernel (EGOMI_GEMM_TILE=8).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M, N, K = 32262, 4096, 1280
a = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
c = torch.zeros(M, N, device="cuda", dtype=torch.float32)
def t(fn, n=10):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for acc in (False, True):
    us = t(lambda: ops.mm(a, w, out=c, accumulate=acc))
    print(f"EGOMI_GEMM_TILE={os.environ.get('EGOMI_GEMM_TILE', '-')} accumulate={acc}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
