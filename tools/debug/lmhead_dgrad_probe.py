#!/usr/bin/env python3
"""lm_head dgrad of the bench step (M = B*(S-Lp) = 1280 span rows, N = 4096, K = V64 = 32320): the 128x128 two-stage kernel
(what tile_choice picks: only 80 tiles of 256x256) vs the 256x256 8-phase kernel with every tile row K-sliced
(EGOMI_GEMM_TILE=8 python tools/debug/lmhead_dgrad_probe.py).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
M, N, K = 1280, 4096, 32320
a = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
ref = a.float() @ w.float().t()
def t(fn, n=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
forced = os.environ.get("EGOMI_GEMM_TILE")
if forced != "8":
    us = t(lambda: ops.mm(a, w, out=c))
    print(f"library choice: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  err {float((c.float()-ref).abs().max()):.2e}")
else:
    rows = (M + 255) // 256
    for S in (2, 3, 4, 5, 6, 8):
        us = t(lambda: ops.mm(a, w, out=c, workspace=ws, split_k=rows * 16 + S, persistent=False))
        print(f"8-phase, all {rows} tile rows in {S} K-slices: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  err {float((c.float()-ref).abs().max()):.2e}")
if forced != "8":
    from egoscaler_amd.ops import _tail_workspace
    us = t(lambda: ops.mm(a, w, out=c, workspace=ws))
    print(f"library choice with workspace (kernel id {ops.gemm_kernel_id(M, N, K)} without one): {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  err {float((c.float()-ref).abs().max()):.2e}")
