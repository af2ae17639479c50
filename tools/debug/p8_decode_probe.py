#!/usr/bin/env python3
"""M = 256 decode projections on the PERSISTENT 256x256 8-phase kernel (stream-K over all CUs, partial tiles summed inside the launch by tickets) instead of the
256 x 128 ring kernel: half the LDS fill bytes per flop.  EGOMI_GEMM_TILE=8 forces the tile family, ws_tickets_zeroed = 2 the persistent form.
  EGOMI_GEMM_TILE=8 python tools/debug/p8_decode_probe.py      GPU box only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops, _lib
L = _lib.lib()
M = 256
ws = torch.zeros((4096 + 256 * 2 * 262144) // 4, dtype=torch.float32, device="cuda")
def launch(a, w, c, pers):
    d = ops.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), w.data_ptr(), c.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = a.shape[0], w.shape[0], a.shape[1], a.stride(0), w.stride(0), c.stride(0)
    d.a_layout, d.b_layout, d.ab_dtype, d.c_dtype, d.batch, d.batch_inner = 0, 0, ops.dt(a.dtype), ops.dt(c.dtype), 1, 1
    d.alpha = 1.0
    if pers:
        d.workspace, d.workspace_bytes, d.ws_tickets_zeroed = ws.data_ptr(), ws.numel() * 4, 2
    ops.call("egomi_gemm", ctypes.byref(d), ops.S())
for N, K in [(22016, 4096), (12288, 4096), (4096, 4096), (4096, 11008), (32262, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    nw = max(3, -(-(800 << 20) // (N * K * 2)))
    wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = a.float() @ wl[0].float().t()
    line = f"M=256 N={N:6d} K={K:6d}:"
    for pers in (True, False):
        launch(a, wl[0], c, pers); torch.cuda.synchronize()
        err = float((c.float() - ref).abs().max() / ref.abs().max())
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8):
                launch(a, wl[(rnd * 8 + i + 1) % nw], c, pers)
            e1.record(); torch.cuda.synchronize()
            if rnd: ts.append(e0.elapsed_time(e1) / 8 * 1e3)
        ts.sort()
        line += f"  {'persistent 256x256' if pers else 'library default   '} {ts[len(ts)//2]:6.1f} us (err {err:.1e})"
    print(line, flush=True)
