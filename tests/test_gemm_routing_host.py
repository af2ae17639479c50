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


def tn_desc(M, N, K, a_layout, c_dtype=BF16, ws=True):
    d = desc(M, N, K)
    d.a_layout, d.b_layout = a_layout, 1
    d.lda = M if a_layout == 1 else K
    d.ldb, d.ldc, d.c_dtype = N, N, c_dtype
    if not ws:
        d.workspace, d.workspace_bytes, d.ws_tickets_zeroed = None, 0, 0
    return d


def test_k_major_products_slice_the_ragged_last_round():
    """The unfrozen step's data gradients at M = 5536, N = 4096 are 352 tiles of 256x256 on 256 CUs (2 rounds for 1.375 rounds of work):
    the k-major kernel K-slices its last tile rows like the K-contiguous one.  Exact multiples of a round and products without scratch
    stay whole; every slice keeps >= 8 K-tiles; the slabs fit the scratch."""
    from egoscaler_amd.ops import F32
    L = _lib.lib()
    row0, sl = ctypes.c_int(0), ctypes.c_int(0)
    for K in (4096, 12288, 22016):                                         # o_proj, [Wq;Wk;Wv], [Wgate;Wup] data gradients
        d = tn_desc(5536, 4096, K, 0)
        assert L.egomi_gemm_kernel_id(ctypes.byref(d)) == 3
        # round 4: by default these take the 352x256 form (16 x 16 tiles = one whole round): no K-sliced rows at all ...
        assert L.egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl)) == 0 and sl.value == 0 and row0.value == 5536
        # ... and with that form switched off the k-major kernel's own 256x256 tiles slice their last rows
        assert L.egomi_gemm_set_tall(0) == 0
        try:
            assert L.egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl)) == 0
        finally:
            L.egomi_gemm_set_tall(-1)
        assert sl.value >= 2 and row0.value % 256 == 0 and 0 <= row0.value < 5536, (K, row0.value, sl.value)
        assert (K // 64) // sl.value >= 8
        assert (5536 - row0.value) * 4096 * 4 * sl.value <= d.workspace_bytes - 4096
        dn = tn_desc(5536, 4096, K, 0, ws=False)
        assert L.egomi_gemm_tn_tail_plan(ctypes.byref(dn), ctypes.byref(row0), ctypes.byref(sl)) == 0 and sl.value == 0 and row0.value == 5536
    d = tn_desc(12288, 4096, 5536, 1, c_dtype=F32)                         # stacked q|k|v weight gradient: 768 tiles = 3 rounds exactly
    assert L.egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl)) == 0 and sl.value == 0
    d = tn_desc(300, 200, 512, 1)                                          # not a product of that kernel
    assert L.egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl)) != 0
