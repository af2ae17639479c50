#!/usr/bin/env python3
"""k-major 8-phase kernel (csrc/gemm_tn.hip) against the K-contiguous 8-phase kernel on pre-transposed operands, at the unfrozen step's
weight-gradient and data-gradient shapes (rows = 5536).  Operands rotate over > 256 MB so that nothing is Infinity-Cache resident.
GPU box only:  python tools/gemm_tn_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

dev = torch.device("cuda")
R = 5536


def timed(fn, reps=12):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


res = []
g = torch.Generator(device=dev).manual_seed(0)
for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
    nb = 6
    dY = [torch.randn(R, N, device=dev, generator=g).bfloat16() for _ in range(nb)]
    X = [torch.randn(R, K, device=dev, generator=g).bfloat16() for _ in range(nb)]
    G = torch.zeros(N, K, device=dev)
    Rp = (R + 63) // 64 * 64
    dYt = [ops.transpose(a, ldo=Rp) for a in dY]
    Xt = [ops.transpose(a, ldo=Rp) for a in X]
    fl = 2.0 * R * N * K
    kid = ops.mm_kernel_id(dY[0], X[0], G, a_layout=1, b_layout=1)
    t_tn = timed(lambda i: ops.mm(dY[i % nb], X[i % nb], out=G, a_layout=1, b_layout=1))
    t_nt = timed(lambda i: ops.mm(dYt[i % nb], Xt[i % nb], out=G))
    t_tr = timed(lambda i: (ops.transpose(dY[i % nb], ldo=Rp, out=dYt[i % nb]), ops.transpose(X[i % nb], ldo=Rp, out=Xt[i % nb])))
    res.append({"wgrad": f"[{N},{K}] += dY^T[{N},{R}] . X[{R},{K}]", "kernel_id": kid, "tn_us": round(t_tn * 1e6, 1), "tn_TFLOPs": round(fl / t_tn / 1e12, 1),
                "nt_us": round(t_nt * 1e6, 1), "nt_TFLOPs": round(fl / t_nt / 1e12, 1), "transposes_us": round(t_tr * 1e6, 1)})
    del dY, X, dYt, Xt, G
    torch.cuda.empty_cache()
for N, K in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)):          # dX[R, K] = dY[R, N] . W[N, K]; the last two: [Wq;Wk;Wv], [Wgate;Wup]
    nb = 6
    dY = [torch.randn(R, N, device=dev, generator=g).bfloat16() for _ in range(nb)]
    W = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(nb)]
    Wt = [ops.transpose(w) for w in W]
    out = torch.zeros(R, K, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * R * N * K
    kid = ops.mm_kernel_id(dY[0], W[0], out, b_layout=1)
    t_nn = timed(lambda i: ops.mm(dY[i % nb], W[i % nb], out=out, b_layout=1))
    t_nt = timed(lambda i: ops.mm(dY[i % nb], Wt[i % nb], out=out))
    res.append({"dgrad": f"[{R},{K}] = dY[{R},{N}] . W[{N},{K}]", "kernel_id": kid, "nn_us": round(t_nn * 1e6, 1), "nn_TFLOPs": round(fl / t_nn / 1e12, 1),
                "nt_us": round(t_nt * 1e6, 1), "nt_TFLOPs": round(fl / t_nt / 1e12, 1)})
    del dY, W, Wt, out
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
