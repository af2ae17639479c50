"""CPU: the host-side routing of egomi_gemm (no kernel is launched, no GPU is touched).

ADVICE r2 (high): egomi_gemm_tail_plan() was evaluated with epilogue = NONE while the launch that follows carries
EGOMI_EPI_SLABS; tile_choice() looks at the epilogue, so for per-GPU M in [1024, 1792] at K >= 8192 (bs = 2 per rank at S = 692:
down_proj, the K-concatenated dgrads) the plan said "256x256 kernel, tail rows K-sliced" and the launch took the 128x128 kernel,
which refuses slabs at M > 512.  The plan and the launch must agree for EVERY shape."""
import ctypes

import pytest

from egoscaler_amd import _lib
from egoscaler_amd.ops import GemmDesc, BF16

# (N, K) of every product the bf16 training step defers a tail on (engine.py: qkv, o_proj, down_proj, qkv / gate|up dgrads)
SHAPES = [(12288, 4096), (4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016)]


def desc(M, N, K, ws_bytes=4096 + 256 * 2 * 262144):
    d = GemmDesc()
    d.A = d.B = d.C = 256
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
    d.ab_dtype, d.c_dtype, d.batch, d.alpha = BF16, BF16, 1, 1.0
    d.workspace, d.workspace_bytes, d.ws_tickets_zeroed = 4096, ws_bytes, 1
    return d


@pytest.mark.parametrize("N,K", SHAPES)
def test_tail_plan_and_launch_pick_the_same_kernel(N, K):
    L = _lib.lib()
    bad = []
    for M in list(range(64, 8192 + 1, 8)) + [1384, 5536, 692, 2768]:
        d = desc(M, N, K)
        row0, sl = ctypes.c_int(0), ctypes.c_int(0)
        rc = L.egomi_gemm_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl))
        if rc == 0 and sl.value >= 2:
            assert 0 <= row0.value < M and row0.value % 256 == 0
            d.epilogue = 2                                   # what ops.gemm_raw sets before egomi_gemm
            if L.egomi_gemm_kernel_id(ctypes.byref(d)) != 2:
                bad.append(M)
    assert not bad, f"plan says 256x256 + slabs, launch would take another kernel at M = {bad[:8]}..."


def test_advice_r2_reproducer_shape():
    """M = 1384 (bs 2 x S 692), down_proj: before the fix plan rc 0 / slices 2 / kernel 2 before and 1 after."""
    L = _lib.lib()
    d = desc(1384, 4096, 11008)
    row0, sl = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.egomi_gemm_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl))
    assert rc != 0 or sl.value < 2 or (setattr(d, "epilogue", 2) or L.egomi_gemm_kernel_id(ctypes.byref(d)) == 2)
    d.epilogue = 0
    assert L.egomi_gemm_kernel_id(ctypes.byref(d)) == 2       # the undeferred call still takes the every-row-sliced 256x256 form
