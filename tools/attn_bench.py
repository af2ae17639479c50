#!/usr/bin/env python3
"""Timing of the fused attention kernels at the bench step's shape (B=8, H=32, S=692, hd=128, causal + key mask).
python tools/attn_bench.py [S] [B]   (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 692
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, hd = 32, 128
M, d = B * S, H * hd
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
dout = (torch.randn(M, d, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
delta = torch.empty_like(lse)
mask = torch.ones(B, S, device="cuda", dtype=torch.uint8)
scale = hd ** -0.5


def timeit(fn, n=10):
    fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]


f = 4.0 * B * H * S * S * hd / 2
from egoscaler_amd import _lib
groups = [int(g) for g in os.environ.get("ATTN_GROUPS", "0").split(",")]
for rnd in range(3):                                       # the forms of the forward kernel, alternated on this box
    for form, grp in [(2, 0)] + [(3, g) for g in groups] + [(4, 0)]:
        _lib.lib().egomi_attn_set_fwd_form(form)
        _lib.lib().egomi_attn_set_fwd_group(grp)
        tf = timeit(lambda: ops.attn_fwd(qkv, B, S, H, hd, scale, out, lse, causal=True, key_mask=mask))
        print(f"B={B} S={S}: fwd form {form} group {grp}: {tf*1e3:7.1f} us {f/tf/1e9:7.1f} TFLOP/s", flush=True)
_lib.lib().egomi_attn_set_fwd_form(4)
for form in (2, 3, 2, 3, 2, 3):
    _lib.lib().egomi_attn_set_bwd_form(form)
    tb = timeit(lambda: ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, scale, causal=True, key_mask=mask))
    print(f"B={B} S={S}: bwd form {form}: {tb*1e3:7.1f} us {2.5*f/tb/1e9:7.1f} TFLOP/s")
print(f"B={B} S={S}: fwd {tf*1e3:7.1f} us {f/tf/1e9:7.1f} TFLOP/s   bwd {tb*1e3:7.1f} us {2.5*f/tb/1e9:7.1f} TFLOP/s (5 products counted)")
