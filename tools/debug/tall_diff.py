#!/usr/bin/env python3
"""Where and by how much the 352x256 form differs from the 256x256 form (no workspace: whole tiles only).  GPU box only."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops, _lib
L = _lib.lib()
torch.manual_seed(0)
M, N, K = 5536, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
outs = []
for mode in (0, 2, 2):
    L.egomi_gemm_set_tall(ctypes.c_int(mode))
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    ops.mm(a, w, out=c)
    torch.cuda.synchronize()
    outs.append(c)
ref = (a.double() @ w.double().t())
d = (outs[0] - outs[1]).abs()
print("tall run-to-run identical:", torch.equal(outs[1], outs[2]))
print("differing elements:", int((d > 0).sum()), "of", M * N, "max diff", float(d.max()), "max |ref|", float(ref.abs().max()))
print("err vs fp64: form256", float((outs[0].double() - ref).abs().max()), "tall", float((outs[1].double() - ref).abs().max()))
idx = (d > 0).nonzero()
if len(idx):
    rows = idx[:, 0]; cols = idx[:, 1]
    print("rows with diffs: min", int(rows.min()), "max", int(rows.max()), "distinct row%352:", sorted(set((rows % 352).tolist()))[:40])
    print("cols%256 distinct:", sorted(set((cols % 256).tolist()))[:40])
    print("hist of diff magnitudes:", torch.histc(d[d > 0].log10(), bins=8, min=-8, max=0).tolist())
