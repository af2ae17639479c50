import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
from egoscaler_amd import ops
dev = torch.device("cuda"); R = 5536
def timed(fn, reps=12):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
g = torch.Generator(device=dev).manual_seed(0)
out = []
for N, K in ((4096, 4096), (11008, 4096)):
    for pad in (0, 64, 72, 136):
        nb = 6
        dY = [torch.randn(R, N + pad, device=dev, generator=g).bfloat16()[:, :N] for _ in range(nb)]
        X = [torch.randn(R, K + pad, device=dev, generator=g).bfloat16()[:, :K] for _ in range(nb)]
        G = torch.zeros(N, K, device=dev)
        fl = 2.0 * R * N * K
        assert ops.mm_kernel_id(dY[0], X[0], G, a_layout=1, b_layout=1) == 3
        t = timed(lambda i: ops.mm(dY[i % nb], X[i % nb], out=G, a_layout=1, b_layout=1))
        out.append({"N": N, "K": K, "pad": pad, "us": round(t * 1e6, 1), "TFLOPs": round(fl / t / 1e12, 1)})
        del dY, X, G; torch.cuda.empty_cache()
print(json.dumps(out))
