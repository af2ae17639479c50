#!/usr/bin/env python3
"""Quick numerical screen of egomi_gemm's tuned path (whatever EGOMI_GEMM_TILE selects) against torch fp32 matmul of the
same bf16 operands, over shapes that exercise ragged M/N, odd / single K-tile counts and the epilogue.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

torch.manual_seed(0)
bad = 0
shapes = [(256, 256, 64), (256, 256, 128), (256, 256, 192), (300, 520, 320), (5536, 4096, 4096), (5536, 1024, 384),
          (1000, 11008, 4096), (4096, 4096, 5568), (77, 200, 448), (513, 257, 704),
          (5536, 12288, 4096), (5536, 4096, 11008), (2900, 4100, 2048)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16()
    ref = a.float() @ w.float().t()
    ref2 = ref + bias.float() + res.float()
    for r in range(reps):
        c = ops.mm(a, w)
        c2 = ops.mm(a, w, bias=bias, residual=res)
        cf = ops.mm(a, w, out_dtype=torch.float32)
        c3 = ops.mm(a, w, out=res.clone(), residual=res, accumulate=True)          # plain residual + accumulate (fast epilogue)
        c4 = ops.mm(a, w, residual=res)
        e4 = max(((c3.float() - (ref + 2 * res.float())).abs().max() / ref2.abs().max()).item(),
                 ((c4.float() - (ref + res.float())).abs().max() / ref2.abs().max()).item())
        e1 = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        e2 = ((c2.float() - ref2).abs().max() / ref2.abs().max()).item()
        e3 = ((cf - ref).abs().max() / ref.abs().max()).item()
        ok = e1 < 1e-2 and e2 < 1e-2 and e3 < 1e-4 and e4 < 1e-2
        if not ok or r == 0:
            print(f"M={M} N={N} K={K}: bf16 {e1:.2e}  bias+res {e2:.2e}  f32-out {e3:.2e}  res/acc {e4:.2e}  {'ok' if ok else 'MISMATCH'}")
        bad += (not ok)
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
