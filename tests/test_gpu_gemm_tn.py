"""GPU: the k-major 8-phase GEMM (csrc/gemm_tn.hip, egomi_gemm_kernel_id == 3) against fp32 torch products of the same bf16 operands:
  * weight-gradient form  C[M,N] (+)= A^T . B  with A [K,M], B [K,N]  (a_layout = b_layout = 1): fp32 output, overwrite and accumulate,
    ragged reduction lengths (K % 64 != 0: the zero-page rows), ragged tile edges (M, N not multiples of 256 / 128 / 8), operands that are
    column slices of wider arrays (dqkv[:, d:2d]), the padded-logits case (ld > N), the bench's own shapes;
  * data-gradient form    C[M,N] (+)= A . B    with A [M,K], B [K,N]   (a_layout = 0, b_layout = 1): bf16 output, with accumulation.
Replaces the matmuls of nn.Linear's backward (train.py:183 for the layers `--unfreeze_language_model` trains, model_arch.py:33-51).
Also: the products the tuned kernel refuses still run (generic kernel), and EGOMI_GEMM_TN-off behaviour is the old route (engine test)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from egoscaler_amd import ops as O
    return O


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).bfloat16()


def rel(got, ref):
    return float((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize("M,N,K,acc", [(4096, 4096, 5536, False), (4096, 4096, 5536, True), (2104, 2072, 1284, False), (4104, 4096, 640, True),
                                       (11008, 4096, 1408, False), (4096, 11008, 1384, True), (2048, 2048, 256, False)])
def test_wgrad_form_fp32_out(ops, M, N, K, acc):
    a, b = rnd(K, M, seed=1), rnd(K, N, seed=2, scale=0.1)
    A, B = a.cuda(), b.cuda()
    C0 = torch.randn(M, N, generator=torch.Generator().manual_seed(3))
    C = C0.clone().cuda() if acc else torch.full((M, N), float("nan"), device="cuda")
    assert ops.mm_kernel_id(A, B, C, a_layout=1, b_layout=1, accumulate=acc) == 3
    ops.mm(A, B, out=C, a_layout=1, b_layout=1, accumulate=acc)
    ref = a.float().t() @ b.float() + (C0 if acc else 0)
    assert rel(C, ref) < 2e-3, rel(C, ref)                     # fp32 accumulation of bf16 products: only the summation order differs


def test_wgrad_form_on_column_slices_and_padded_rows(ops):
    """dY = a column block of a wider array (dqkv[:, d:2d]: lda = 3d); the lm_head case: A = padded logits [rows, V64] used up to V."""
    K, d = 1384, 2048
    wide = rnd(K, 3 * d, seed=5).cuda()
    X = rnd(K, 2304, seed=6, scale=0.1).cuda()
    A = wide[:, d:2 * d]
    C = torch.zeros(d, 2304, device="cuda")
    assert ops.mm_kernel_id(A, X, C, a_layout=1, b_layout=1) == 3
    ops.mm(A, X, out=C, a_layout=1, b_layout=1)
    assert rel(C, A.float().cpu().t() @ X.float().cpu()) < 2e-3
    V, Vp, rows, dm = 16390, 16448, 1280, 1024                   # V % 8 != 0, rows of the array are Vp wide
    lg = rnd(rows, Vp, seed=7).cuda()
    hn = rnd(rows, dm, seed=8, scale=0.1).cuda()
    Av = lg[:, :V]
    G = torch.zeros(V, dm, device="cuda")
    assert ops.mm_kernel_id(Av, hn, G, a_layout=1, b_layout=1) == 3
    ops.mm(Av, hn, out=G, a_layout=1, b_layout=1)
    assert rel(G, Av.float().cpu().t() @ hn.float().cpu()) < 2e-3


@pytest.mark.parametrize("M,N,K,acc", [(5536, 4096, 4096, False), (5536, 4096, 4096, True), (1384, 4096, 11008, False), (2050, 4104, 1024, True)])
def test_dgrad_form_bf16_out(ops, M, N, K, acc):
    """dX = dY . W with W [K_red, N] as the nn.Linear weight lies in memory; accumulate = the k / v contributions added into d_h."""
    a, w = rnd(M, K, seed=11), rnd(K, N, seed=12, scale=0.05)
    A, W = a.cuda(), w.cuda()
    C0 = rnd(M, N, seed=13)
    C = C0.clone().cuda() if acc else torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    assert ops.mm_kernel_id(A, W, C, b_layout=1, accumulate=acc) == 3
    ops.mm(A, W, out=C, b_layout=1, accumulate=acc)
    ref = a.float() @ w.float() + (C0.float() if acc else 0)
    assert rel(C, ref) < 2e-2, rel(C, ref)


@pytest.mark.parametrize("form,M,N,K,acc", [("dgrad", 5536, 4096, 12288, False), ("dgrad", 5536, 4096, 22016, False), ("dgrad", 5536, 4096, 4096, True),
                                            ("wgrad", 22016, 4096, 5536, True), ("wgrad", 4608, 4096, 5532, False), ("wgrad", 2048, 11008, 5532, True)])
def test_k_sliced_tail_rows(ops, form, M, N, K, acc, request):
    """Ragged last round: the last tile rows are computed as K-slices (fp32 slabs in the scratch) and summed by the combine pass
    (egomi_gemm_tn_tail_plan says which rows).  The sliced rows must be as good as the whole ones: compared separately, with the ragged
    reduction tail (K % 64 != 0) inside the last slice, accumulation into C on both kinds of rows, bf16 and fp32 outputs."""
    import ctypes
    from egoscaler_amd import _lib
    # (this kernel's own 256x256 tiles: since round 4 a plain data gradient of these shapes would otherwise take the 352x256 form, which has no K-sliced rows —
    #  tests/test_gpu_gemm_tall.py)
    _lib.lib().egomi_gemm_set_tall(ctypes.c_int(0))
    request.addfinalizer(lambda: _lib.lib().egomi_gemm_set_tall(ctypes.c_int(-1)))
    wg = form == "wgrad"
    a = rnd(K, M, seed=31) if wg else rnd(M, K, seed=31)
    b = rnd(K, N, seed=32, scale=0.05)
    A, B = a.cuda(), b.cuda()
    odt = torch.float32 if wg else torch.bfloat16
    C0 = torch.randn(M, N, generator=torch.Generator().manual_seed(33)).to(odt)
    C = C0.clone().cuda() if acc else torch.full((M, N), float("nan"), dtype=odt, device="cuda")
    al = 1 if wg else 0
    assert ops.mm_kernel_id(A, B, C, a_layout=al, b_layout=1, accumulate=acc) == 3
    d = ops.GemmDesc()
    d.A, d.B, d.C, d.M, d.N, d.K = A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K
    d.lda, d.ldb, d.ldc, d.a_layout, d.b_layout = A.stride(0), N, N, al, 1
    d.ab_dtype, d.c_dtype, d.batch, d.alpha, d.accumulate = ops.BF16, ops.dt(odt), 1, 1.0, int(acc)
    ws = ops._tail_workspace(A.device)
    d.workspace, d.workspace_bytes, d.ws_tickets_zeroed = ws.data_ptr(), ws.numel() * 4, 1
    row0, sl = ctypes.c_int(0), ctypes.c_int(0)
    assert _lib.lib().egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(sl)) == 0
    assert sl.value >= 2 and 0 <= row0.value < M, (row0.value, sl.value)          # these shapes were chosen because the plan slices them
    ops.mm(A, B, out=C, a_layout=al, b_layout=1, accumulate=acc)
    ref = (a.float().t() if wg else a.float()) @ b.float() + (C0.float() if acc else 0)
    tol = 2e-3 if wg else 2e-2
    r0 = row0.value
    assert rel(C[:r0], ref[:r0]) < tol and rel(C[r0:], ref[r0:]) < tol, (rel(C[:r0], ref[:r0]), rel(C[r0:], ref[r0:]))
    assert float(ws[:1024].abs().max()) == 0                                       # the ticket words ahead of the slabs stay zero


def test_products_the_kernel_refuses_still_run(ops):
    a, b = rnd(300, 200, seed=21).cuda(), rnd(300, 136, seed=22).cuda()          # few tiles
    C = torch.zeros(200, 136, device="cuda")
    assert ops.mm_kernel_id(a, b, C, a_layout=1, b_layout=1) == 0
    ops.mm(a, b, out=C, a_layout=1, b_layout=1)
    assert rel(C, a.float().cpu().t() @ b.float().cpu()) < 2e-3
    a2, b2 = rnd(1386, 4096, seed=23).cuda(), rnd(1386, 4096, seed=24).cuda()    # K % 4 != 0: a 4-row DMA piece would straddle the end
    C2 = torch.zeros(4096, 4096, device="cuda")
    assert ops.mm_kernel_id(a2, b2, C2, a_layout=1, b_layout=1) == 0
    ops.mm(a2, b2, out=C2, a_layout=1, b_layout=1)
    assert rel(C2, a2.float().cpu().t() @ b2.float().cpu()) < 2e-3
