#!/usr/bin/env python3
"""352x256 form of the 8-phase GEMM (gemm_nt_bf16_tall_kernel) vs the 256x256 form: correctness against a torch fp32 product and cold-weight
timing per shape.  The form is chosen per process (EGOMI_GEMM_TALL=0|1|2 is read once):  EGOMI_GEMM_TALL=2 python tools/debug/tall_probe.py [check]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops, _lib
if os.environ.get("EGOMI_LIB"):                      # a variant build of the library (e.g. -DTL_SCHED=3)
    _lib.LIB_PATH = os.environ["EGOMI_LIB"]

mode = os.environ.get("EGOMI_GEMM_TALL", "1")
if "check" in sys.argv:
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K, kw) in [(5536, 4096, 4096, {}), (5536, 4096, 2048, {}), (5000, 4104, 2112, {}), (352, 256, 2048, {}), (360, 264, 2048, {}),
                          (5536, 4096, 4096, {"res": 1}), (5536, 4096, 4096, {"acc": 1}), (5536, 4096, 4096, {"f32": 1}), (1408, 512, 4160, {}), (8192, 4096, 4096, {})]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        ref = a.float() @ w.float().t()
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if kw.get("f32") else torch.bfloat16)
        args = {}
        if kw.get("res"):
            r = torch.randn(M, N, device="cuda").bfloat16(); args["residual"] = r; ref = ref + r.float()
        if kw.get("acc"):
            out.copy_(torch.randn(M, N, device="cuda").bfloat16()); ref = ref + out.float(); args["accumulate"] = True
        ops.mm(a, w, out=out, **args)
        err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, err)
        print(f"TALL={mode} M={M} N={N} K={K} {kw}: rel err {err:.2e}", flush=True)
    assert worst < 1.5e-2, worst
    print("check ok")
    sys.exit(0)

M = 5536
ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
for N, K in [(4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (12288, 4096), (11008, 4096), (22016, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    nw = max(2, -(-(640 << 20) // (N * K * 2)))
    wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ts = []
    for rnd in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.mm(a, wl[0], out=c, workspace=ws)
        e0.record()
        for i in range(8):
            ops.mm(a, wl[(rnd * 8 + i + 1) % nw], out=c, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1) / 8)
    ts.sort(); med = ts[len(ts) // 2]
    print(f"TALL={mode} M={M} N={N:6d} K={K:6d}: {med*1e3:7.1f} us  {2*M*N*K/med/1e9:7.1f} TFLOP/s (best {2*M*N*K/ts[0]/1e9:.1f})", flush=True)
