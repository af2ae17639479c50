"""Few launches of the k-major kernel (id 3) and of the K-contiguous kernel (id 2) at one weight-gradient shape, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import ops
dev = torch.device("cuda"); R, N, K = 5536, 4096, 4096
g = torch.Generator(device=dev).manual_seed(0)
dY = torch.randn(R, N, device=dev, generator=g).bfloat16()
X = torch.randn(R, K, device=dev, generator=g).bfloat16()
G = torch.zeros(N, K, device=dev)
Rp = (R + 63) // 64 * 64
dYt, Xt = ops.transpose(dY, ldo=Rp), ops.transpose(X, ldo=Rp)
for _ in range(4):
    ops.mm(dY, X, out=G, a_layout=1, b_layout=1)
    ops.mm(dYt, Xt, out=G)
torch.cuda.synchronize()
