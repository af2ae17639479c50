#!/usr/bin/env python3
"""Where a block of the per-tile 256x256 GEMM kernel spends its life: needs a library built with -DGEMM_STAMP on
egoscaler_amd/csrc/gemm_fast.hip (s_memtime stamps, wave 0 of every block, plain-bf16 epilogue path) in place of
egoscaler_amd/lib/libegomi.so.  python tools/debug/gemm_stamp.py [M N K]   (GPU box only)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops, _lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 4096)
a = torch.randn(M, K, device="cuda").bfloat16()
nw = max(1, -(-(640 << 20) // (N * K * 2)))
ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]          # rotating weights: cold, as in the step
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
L = ctypes.CDLL(_lib.LIB_PATH)
for i in range(3):
    ops.mm(a, ws[i % nw], out=c)
torch.cuda.synchronize()
L.egomi_gemm_stamp_reset()
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    ops.mm(a, ws[(i + 3) % nw], out=c)
e1.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.egomi_gemm_stamp_read(buf)
names = ["set-up + DMA issue", "wait first tile", "main loop", "drain + barrier", "epilogue issue", "stores acknowledged"]
blocks = buf[8]
tot = sum(buf[i] for i in range(6))
print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch, {blocks // n} stamped blocks per launch, {tot / blocks:.0f} ticks per block")
for i, nme in enumerate(names):
    print(f"  {nme:22s} {buf[i] / blocks:9.0f} ticks  {100.0 * buf[i] / tot:5.1f} %")
