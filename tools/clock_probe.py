#!/usr/bin/env python3
"""Samples rocm-smi (sclk, power) while egomi_gemm runs back to back at the bench's largest shape: what engine clock the
dense-MFMA kernel actually gets on this box (the 2.5 PFLOP/s peak assumes 2.4 GHz).  GPU box only."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import ops

M, N, K = 5536, 22016, 4096
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stop = False


def spin():
    while not stop:
        for _ in range(50):
            ops.mm(a, w, out=c)
        torch.cuda.synchronize()


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    return [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]


print("idle:", smi())
th = threading.Thread(target=spin)
th.start()
time.sleep(3.0)
for i in range(3):
    print(f"under GEMM load ({i}):", smi())
    time.sleep(1.0)
stop = True
th.join()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.mm(a, w, out=c)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"sustained: {2*M*N*K/t/1e9:.1f} TFLOP/s")
