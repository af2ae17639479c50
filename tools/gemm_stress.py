#!/usr/bin/env python3
"""Randomised screen of egomi_gemm against torch fp32 matmul: shapes around the 8-phase kernel's selection rule, ragged M / N,
odd K-tile counts, every epilogue form, repeated launches (a staged-buffer race shows up as a rare wrong tile).  GPU box only.
python tools/gemm_stress.py [n_shapes] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
torch.manual_seed(seed)
bad = 0
for it in range(n_shapes):
    # (round 4: multiples of 8 around the 352x256 form's selection rule and the column split between the two tile forms — N = 11008 / 22016 at M ~ 5536)
    M = rng.choice([rng.randint(1, 300), rng.randint(300, 3000), rng.randint(3000, 6500), 5536, 4096, rng.randint(44, 800) * 8, 5544, 2816])
    N = rng.choice([rng.randint(8, 600), rng.randint(600, 5000), rng.randint(5000, 13000), 4096, 12288, 11008, 22016, rng.randint(4, 60) * 256]) // 8 * 8
    K = rng.choice([64, 128, 192, 2048, 2112, 4096, rng.randint(1, 100) * 64, rng.randint(32, 172) * 64])
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    ref = a.float() @ w.float().t()
    scale = float(ref.abs().max()) + 1e-6
    res = torch.randn(M, N, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    worst = 0.0
    for rep in range(3):
        c1 = ops.mm(a, w)
        c2 = ops.mm(a, w, residual=res)
        c3 = ops.mm(a, w, bias=bias, act=ops.ACT_GELU)
        c4 = ops.mm(a, w, out=torch.zeros(M, N, device="cuda", dtype=torch.float32), accumulate=True)
        e = max(float((c1.float() - ref).abs().max()) / scale,
                float((c2.float() - (ref + res.float())).abs().max()) / (scale + float(res.float().abs().max())),
                float((c3.float() - torch.nn.functional.gelu(ref + bias.float())).abs().max()) / (scale + 4.0),
                float((c4 - ref).abs().max()) / scale * 20)        # fp32 output: 20x tighter
        worst = max(worst, e)
    ok = worst < 1.5e-2
    bad += (not ok)
    if not ok or it % 10 == 0:
        print(f"[{it}] M={M} N={N} K={K}: worst rel err {worst:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("FAILED" if bad else f"all {n_shapes} shapes ok")
sys.exit(1 if bad else 0)
