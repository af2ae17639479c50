#!/usr/bin/env python3
"""Where a block of attn_fwd3_kernel spends its life.  Builds stamped copies of the library (-DATTN_STAMP -DATTN_STAMP_WAVE=w:
s_memtime stamps, lane 0 of wave w of every block) beside the real one and runs the bench shape through each.
`python tools/debug/attn_stamp3.py build` (anywhere hipcc is) then `python tools/debug/attn_stamp3.py` (GPU box)."""
import ctypes, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "egoscaler_amd", "lib")
WAVES = (0, 3)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from egoscaler_amd import build
    build.build()
    objs = [os.path.join(LIBDIR, f[:-4] + ".o") for f in build.sources() if f != "attention.hip"]
    for w in WAVES:
        o = os.path.join(LIBDIR, f"attention_stamp{w}.o")
        subprocess.check_call([build._hipcc(), *build.COMMON, "-DATTN_STAMP", f"-DATTN_STAMP_WAVE={w}", "-c", os.path.join(build.CSRC, "attention.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "-shared", "-fPIC", f"--offload-arch={build.ARCH}", *objs, o, "-o", os.path.join(LIBDIR, f"libegomi_stamp{w}.so")])
    sys.exit(0)
w = int(os.environ.get("STAMP_WAVE", "3"))
from egoscaler_amd import _lib
_lib.LIB_PATH = os.path.join(LIBDIR, f"libegomi_stamp{w}.so")
import torch
from egoscaler_amd import ops
S, B, H, hd = 692, 8, 32, 128
M, d = B * S, H * hd
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * d, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
mask = torch.ones(B, S, device="cuda", dtype=torch.uint8)
L = _lib.lib()
L.egomi_attn_set_fwd_form(3)                       # the stamps sit in attn_fwd3_kernel (attn_fwd4_kernel runs the same steps)
BWD = "bwd" in sys.argv[1:]
dout = (torch.randn(M, d, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty_like(lse)
ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=mask)
run = (lambda: ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=True, key_mask=mask)) if BWD else \
      (lambda: ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=mask))
for _ in range(5):
    run()
torch.cuda.synchronize()
L.egomi_attn_stamp_reset()
N = 10
for _ in range(N):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.egomi_attn_stamp_read(buf)
names = ["entry: block map, offsets, DMA 0/1 issue", "Q + mask loads issued and landed", "mask commit, vmcnt(0), barrier", "QK(0) + max(0)",
         "top: vmcnt + barrier", "steps", "top: request (4 LDS-DMA) [fwd3] / dead tops [dq3]", "epilogue"]
tot = sum(buf[i] for i in range(8))
blocks = buf[8]
print(f"{'attn_bwd_dq3_kernel' if BWD else 'attn_fwd3_kernel'} wave {w}: blocks stamped {blocks}, cycles per block {tot / blocks:.0f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {buf[i] / blocks:9.0f} cycles/block  {100.0 * buf[i] / tot:5.1f} %")
