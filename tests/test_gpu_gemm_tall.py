"""GPU: the 352x256 form of the 8-phase bf16 GEMM (gemm_nt_bf16_tall_kernel, csrc/gemm_fast.hip; egomi_gemm_set_tall) — the nn.Linear products of the
LLaMA layers the reference runs through torch (modeling_llama.py:150-176 MLP, :216-290 attention projections; backward by autograd, train.py:183).
  * against the 256x256 form on the same operands WITHOUT K-sliced tail rows (split_k = 1): every element accumulates its K-tiles in the same order in
    both forms, so the results must be the same bits — plain, + residual, accumulate, fp32 output, ragged M / N (multiples of 8);
  * against an fp32 torch product (the tolerance of the other bf16 GEMM tests);
  * every operand ending exactly at the end of an allocation of its own, aligned (M % 352 == 0, N % 256 == 0) and ragged (test_gpu_bounds.py's placement);
  * the library's own choice at the bench step's shapes (M = 5536): N = 4096 and N = 12288 take the tall form (no tail rows), N = 22016 does not."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from egoscaler_amd import ops as O
    return O


@pytest.fixture()
def tall(ops):
    from egoscaler_amd import _lib
    L = _lib.lib()

    def set_mode(m):
        assert L.egomi_gemm_set_tall(ctypes.c_int(m)) == 0
    yield set_mode
    L.egomi_gemm_set_tall(ctypes.c_int(-1))


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


CASES = [(5536, 4096, 4096, "plain"), (5536, 4096, 2048, "residual"), (5536, 4096, 2112, "accumulate"), (5536, 4096, 2048, "f32"),
         (5000, 4104, 2112, "plain"), (5000, 4104, 2112, "residual"), (352, 256, 2048, "plain"), (2824, 4104, 2048, "residual"),
         (2816, 4096, 4160, "plain"), (8192, 4096, 2048, "plain"), (5536, 12288, 2048, "plain"), (3168, 2048, 8192, "f32_residual")]


@pytest.mark.parametrize("M,N,K,kind", CASES)
def test_tall_form_is_bit_identical_to_the_256_form_and_close_to_fp32(ops, tall, M, N, K, kind):
    a, w = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=0.05).cuda()
    odt = torch.float32 if kind.startswith("f32") else torch.bfloat16
    r = rnd(M, N, seed=3, dtype=odt).cuda() if "residual" in kind else None
    c0 = rnd(M, N, seed=4, dtype=odt).cuda() if kind == "accumulate" else torch.full((M, N), 7.0, dtype=odt, device="cuda")
    tall(0)
    big = ops.mm_kernel_id(a, w, c0, accumulate=kind == "accumulate") == 2      # smaller products take the 128x128 kernel in either mode
    out = []
    for mode in (0, 2):
        tall(mode)
        c = c0.clone()
        kw = {}
        if r is not None:
            kw["residual"] = r
        if kind == "accumulate":
            kw["accumulate"] = True
        ops.mm(a, w, out=c, split_k=1, **kw)                      # split_k = 1: the 256x256 form runs whole tiles only (no K-sliced tail rows)
        torch.cuda.synchronize()
        out.append(c)
    ref = a.float() @ w.float().t()
    if r is not None:
        ref += r.float()
    if kind == "accumulate":
        ref += c0.float()
    assert float((out[1].float() - ref).abs().max()) <= (2e-2 if odt == torch.bfloat16 else 1e-4) * float(ref.abs().max())
    assert torch.equal(out[0], out[1])
    assert big or M * N < 128 * 65536


@pytest.mark.parametrize("M,N,K", [(5632, 4096, 2048), (5536, 4096, 2112), (3520, 4096, 4096), (2816, 4096, 2048), (2824, 4104, 2112)])
def test_tall_form_operands_at_the_end_of_their_allocations(ops, tall, M, N, K):
    from tests.test_gpu_bounds import at_end
    tall(2)
    a0, w0, r0 = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=0.05).cuda(), rnd(M, N, seed=5).cuda()
    for with_res in (False, True):
        res = []
        for placed in (False, True):
            keep = []
            put = (lambda t: at_end(t, keep)) if placed else (lambda t: t.clone())
            A, W = put(a0), put(w0)
            R = put(r0) if with_res else None
            C = put(torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda"))
            ops.mm(A, W, out=C, **({"residual": R} if with_res else {}))
            torch.cuda.synchronize()
            res.append(C.clone())
            del keep
        assert torch.equal(res[0], res[1])
        ref = a0.float() @ w0.float().t() + (r0.float() if with_res else 0)
        assert float((res[1].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_library_choice_at_the_step_shapes(ops, tall):
    """Default mode at M = 5536: the N = 4096 products (o_proj, down_proj, three data gradients) and q|k|v take the tall form — mm(defer_tail=True)
    reports no pending tail rows, where the 256x256 form (mode 0) leaves K-sliced rows for the next kernel."""
    M = 5536
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    a = rnd(M, 4096, seed=1).cuda()
    for N in (4096, 12288):
        w = rnd(N, 4096, seed=2, scale=0.05).cuda()
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        tall(0)
        _, t0 = ops.mm(a, w, out=c, workspace=ws, defer_tail=True)
        assert t0 is not None and t0[1] >= 2 and 0 < t0[0] < M, (N, t0)
        tall(1)
        _, t1 = ops.mm(a, w, out=c, workspace=ws, defer_tail=True)
        torch.cuda.synchronize()
        assert t1 is None, (N, t1)
        ref = a.float() @ w.float().t()
        assert float((c.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (4096, 2048, 12288), (2048, 4096, 2112)])
def test_tall_form_with_a_k_major_weight_is_bit_identical_to_the_k_major_256_form(ops, tall, M, N, K):
    """Data gradient dX = dY . W against the weight as it lies in memory (b_layout = 1: gemm_tn.hip's product, kernel id 3): the 352x256 form with the
    k-major B side (gemm_nt_bf16_tall_kernel<bf16, true>) against gemm_bf16_8phase_t_kernel on shapes whose 256x256 tiles are whole rounds (no K-sliced
    rows in either): the same bits, and close to fp32."""
    dy, w = rnd(M, K, seed=11).cuda(), rnd(K, N, seed=12, scale=0.05).cuda()
    out = []
    for mode in (0, 2):
        tall(mode)
        c = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
        assert ops.mm_kernel_id(dy, w, c, b_layout=1) == 3
        ops.mm(dy, w, out=c, b_layout=1)
        torch.cuda.synchronize()
        out.append(c)
    ref = dy.float() @ w.float()
    assert float((out[1].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert torch.equal(out[0], out[1])


def test_k_major_data_gradient_at_the_step_shape_takes_the_tall_form(ops, tall):
    """M = 5536, N = 4096 (the q|k|v, o and gate|up data gradients of a trainable layer): default mode reports no K-sliced rows (egomi_gemm_tn_tail_plan),
    mode 0 does; the result is close to fp32 either way and the operands may end an allocation (test_gpu_bounds.py covers the placement)."""
    import ctypes
    from egoscaler_amd import _lib
    M, N, K = 5536, 4096, 4096
    dy, w = rnd(M, K, seed=21).cuda(), rnd(K, N, seed=22, scale=0.05).cuda()
    ref = dy.float() @ w.float()
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    for mode, sliced in ((0, True), (1, False)):
        tall(mode)
        c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        d = ops.GemmDesc()
        d.A, d.B, d.C = dy.data_ptr(), w.data_ptr(), c.data_ptr()
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, N, N
        d.a_layout, d.b_layout, d.ab_dtype, d.c_dtype, d.batch, d.batch_inner, d.alpha = 0, 1, ops.dt(dy.dtype), ops.dt(c.dtype), 1, 1, 1.0
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        row0, slices = ctypes.c_int(0), ctypes.c_int(0)
        assert _lib.lib().egomi_gemm_tn_tail_plan(ctypes.byref(d), ctypes.byref(row0), ctypes.byref(slices)) == 0
        assert (slices.value >= 2) == sliced, (mode, row0.value, slices.value)
        ops.mm(dy, w, out=c, b_layout=1, workspace=ws)
        torch.cuda.synchronize()
        assert float((c.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
