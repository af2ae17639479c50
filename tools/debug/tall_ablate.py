#!/usr/bin/env python3
"""Timing-only ablations of gemm_nt_bf16_tall_kernel (csrc/gemm_fast.hip, -DTL_ABL=mask: 1 no fragment reads, 2 no LDS-DMA, 4 one barrier per phase,
8 no MFMA, 16 every DMA re-reads K-tile 0 (L2-hot operands); results are garbage, only the launch time means anything).  `python tools/debug/tall_ablate.py build` (anywhere hipcc is), then on the GPU box
`python tools/debug/tall_ablate.py` runs every variant in a process of its own on two shapes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "egoscaler_amd", "lib")
MASKS = (0, 2, 16, 24)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from egoscaler_amd import build
    build.build()
    objs = [os.path.join(LIBDIR, f[:-4] + ".o") for f in build.sources() if f != "gemm_fast.hip"]
    for m in MASKS:
        o = os.path.join(LIBDIR, f"gemm_fast_abl{m}.o")
        subprocess.check_call([build._hipcc(), *build.COMMON, f"-DTL_ABL={m}", "-c", os.path.join(build.CSRC, "gemm_fast.hip"), "-o", o])
        subprocess.check_call([build._hipcc(), "-shared", "-fPIC", f"--offload-arch={build.ARCH}", *objs, o, "-o", os.path.join(LIBDIR, f"libegomi_abl{m}.so")])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    m = int(sys.argv[2])
    from egoscaler_amd import _lib
    _lib.LIB_PATH = os.path.join(LIBDIR, f"libegomi_abl{m}.so")
    import ctypes, torch
    from egoscaler_amd import ops
    _lib.lib().egomi_gemm_set_tall(ctypes.c_int(2))
    M = 5536
    line = f"TL_ABL={m:2d}:"
    for N, K in [(4096, 4096), (4096, 12288)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        nw = max(2, -(-(640 << 20) // (N * K * 2)))
        wl = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(nw)]
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.mm(a, wl[0], out=c)
            e0.record()
            for i in range(8):
                ops.mm(a, wl[(rnd * 8 + i + 1) % nw], out=c)
            e1.record(); torch.cuda.synchronize()
            if rnd: ts.append(e0.elapsed_time(e1) / 8)
        ts.sort()
        line += f"  N={N} K={K}: {ts[len(ts) // 2] * 1e3:7.1f} us"
    print(line, flush=True)
    sys.exit(0)
for m in MASKS:
    subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(m)], timeout=120)
