#!/usr/bin/env python3
"""Row kernels of the training step at its own size (M = 5536 rows x 4096 columns, bf16): RMSNorm forward / backward (plain), with
HIP events, achieved GB/s against the algorithmic bytes (one read of every input, one write of every output).  GPU box only.
A/B: EGOMI_RMS_WAVE=0 python tools/bench_rows.py   (block-per-row kernels)  vs  default (wave-per-row for tall bf16 inputs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

M, d = 5536, 4096
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
bufs = [torch.randn(M, d, device=dev, generator=g).bfloat16() for _ in range(24)]      # rotate over > 256 MB: no Infinity-Cache help
w = torch.ones(d, device=dev).bfloat16()
rstd = torch.rand(M, device=dev) + 0.5


def timed(fn, reps=40):
    for i in range(4):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


res = {"rms_wave": os.environ.get("EGOMI_RMS_WAVE", "1")}
n = len(bufs)
t = timed(lambda i: ops.rmsnorm(bufs[(3 * i) % n], w, 1e-6, rstd=rstd, out=bufs[(3 * i + 1) % n]))
res["rmsnorm_fwd"] = {"us": round(t * 1e6, 1), "GBps": round(2 * M * d * 2 / t / 1e9, 1)}
t = timed(lambda i: ops.rmsnorm_bwd(bufs[(4 * i) % n], bufs[(4 * i + 1) % n], w, rstd, dx_add=bufs[(4 * i + 2) % n], out=bufs[(4 * i + 3) % n]))
res["rmsnorm_bwd_with_residual_grad"] = {"us": round(t * 1e6, 1), "GBps": round(4 * M * d * 2 / t / 1e9, 1)}
t = timed(lambda i: ops.rmsnorm_bwd(bufs[(3 * i) % n], bufs[(3 * i + 1) % n], w, rstd, out=bufs[(3 * i + 2) % n]))
res["rmsnorm_bwd"] = {"us": round(t * 1e6, 1), "GBps": round(3 * M * d * 2 / t / 1e9, 1)}
print(json.dumps(res))
