"""GPU: every global operand of the LDS-DMA kernel families placed so that it ENDS exactly at the end of an allocation of its own, at ALIGNED
sizes (S % 32 == 0, M % 256 == 0, K % 64 == 0) as well as ragged ones.  Round 3's page fault hid for two rounds behind
"every test size was ragged": a clamp that only a ragged last tile takes leaves the aligned fast path free to read past the last row.  A read
past such an operand leaves its mapped segment (a fault), and results are compared bit for bit with the same call on ordinary allocations
(a stray read that lands in mapped memory but feeds a result shows there).  Families: fused attention forward / backward (q|k|v, out, dout,
dq|dk|dv, LSE, delta, key mask), the K-contiguous GEMMs (256x256 8-phase, 128x128, the M <= 256 ring kernel, the M <= 16 weight-streaming kernel: A, B, C), the k-major 8-phase
GEMM (weight- and data-gradient forms, a column-sliced operand whose width is not a multiple of 256: ADVICE r3), the decode attention's K/V
cache.  One parametrised test per family; run once."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SEG = 2 << 20


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from egoscaler_amd import ops as O
    return O


def at_end(src: torch.Tensor, keep: list) -> torch.Tensor:
    """A copy of `src` (any shape, contiguous) whose last byte is the last byte of a device allocation of its own: >= 16 MB and a multiple of
    2 MB, requested from an EMPTY cache so that the caching allocator maps a fresh segment of exactly that size instead of splitting a larger
    cached block."""
    n, es = src.numel(), src.element_size()
    nbytes = max(16 << 20, -(-n * es // SEG) * SEG)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    buf = torch.empty(nbytes // es, dtype=src.dtype, device="cuda")
    keep.append(buf)
    t = buf[buf.numel() - n:].view(src.shape)
    t.copy_(src)
    return t


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("S", [128, 256, 512, 100, 200, 692])
@pytest.mark.parametrize("causal", [True, False])
def test_attention_operands_at_the_end_of_their_allocations(ops, S, causal):
    B, H, hd = 2, 2, 128
    d = H * hd
    qkv0 = rnd(B * S, 3 * d, seed=S).cuda()
    dout0 = rnd(B * S, d, seed=S + 1, scale=0.1).cuda()
    km0 = torch.ones(B, S, dtype=torch.uint8, device="cuda")
    km0[-1, S - S // 5:] = 0
    cos, sin = ops.rope_tables(max(S, 8), hd, 10000.0)
    res = []
    for placed in (False, True):
        keep = []
        put = (lambda t: at_end(t, keep)) if placed else (lambda t: t.clone())
        qkv, dout, km = put(qkv0), put(dout0), put(km0)
        out = put(torch.full((B * S, d), 7.0, dtype=torch.bfloat16, device="cuda"))
        lse = put(torch.full((B, H, S), 7.0, dtype=torch.float32, device="cuda"))
        delta = put(torch.full((B, H, S), 7.0, dtype=torch.float32, device="cuda"))
        dqkv = put(torch.full((B * S, 3 * d), 3.0, dtype=torch.bfloat16, device="cuda"))
        rp = (put(cos.cuda()), put(sin.cuda()))
        ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=km)
        ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=causal, key_mask=km, rope=rp)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (out, lse, delta, dqkv)])
        del keep
    for a, b, nm in zip(res[0], res[1], ("out", "lse", "delta", "dqkv")):
        assert torch.equal(a, b), nm
    assert bool(torch.isfinite(res[1][3].float()).all())


# (M, N, K): the 256x256 8-phase kernel (>= 128 tiles, K >= 2048) aligned and ragged; the 128x128 kernel; the M <= 256 ring kernel
@pytest.mark.parametrize("M,N,K,kid", [(5632, 4096, 2048, 2), (5536, 4096, 2112, 2), (5632, 2048, 4096, 2), (1024, 1024, 512, 1), (1000, 1032, 576, 1),
                                       (256, 4096, 4096, None), (256, 12288, 2048, None), (200, 8200, 2112, None),
                                       (8, 4096, 4096, None), (16, 12288, 2048, None), (5, 4100, 1152, None)])    # the last three: gemv_m16_kernel (M <= 16)
def test_k_contiguous_gemm_operands_at_the_end_of_their_allocations(ops, M, N, K, kid):
    a0, w0 = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=0.05).cuda()
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    res = []
    for placed in (False, True):
        keep = []
        put = (lambda t: at_end(t, keep)) if placed else (lambda t: t.clone())
        A, W = put(a0), put(w0)
        C = put(torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda"))
        if kid is not None:
            assert ops.mm_kernel_id(A, W, C) == kid
        ops.mm(A, W, out=C, workspace=ws if M > 512 else None)
        torch.cuda.synchronize()
        res.append(C.clone())
        del keep
    assert torch.equal(res[0], res[1])
    ref = a0.float() @ w0.float().t()
    assert float((res[1].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("form,M,N,K", [("wgrad", 4096, 4096, 5632), ("wgrad", 4096, 4096, 5536), ("wgrad", 2048, 2048, 256), ("dgrad", 5632, 4096, 4096),
                                        ("dgrad", 5536, 4096, 4160), ("wgrad_sliced", 2880, 4096, 5632), ("wgrad_sliced", 2880, 4096, 5536)])
def test_k_major_gemm_operands_at_the_end_of_their_allocations(ops, form, M, N, K):
    """wgrad: C[M,N] = A^T.B, A [K,M], B [K,N];  dgrad: C[M,N] = A.B, A [M,K], B [K,N];  wgrad_sliced: A is the LAST column block (width 2880,
    not a multiple of 256) of a wider [K, 2880 + 2880] array, so the block's last row ends the allocation (engine._wgrad_stacked's fallback)."""
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    if form == "dgrad":
        a0 = rnd(M, K, seed=3).cuda()
    elif form == "wgrad":
        a0 = rnd(K, M, seed=3).cuda()
    else:
        a0 = rnd(K, 2 * M, seed=3).cuda()
    b0 = rnd(K, N, seed=4, scale=0.05).cuda()
    odt = torch.bfloat16 if form == "dgrad" else torch.float32
    res = []
    for placed in (False, True):
        keep = []
        put = (lambda t: at_end(t, keep)) if placed else (lambda t: t.clone())
        A, Bm = put(a0), put(b0)
        if form == "wgrad_sliced":
            A = A[:, M:]
        C = put(torch.full((M, N), 5.0, dtype=odt, device="cuda"))
        al = 0 if form == "dgrad" else 1
        assert ops.mm_kernel_id(A, Bm, C, a_layout=al, b_layout=1) == 3
        ops.mm(A, Bm, out=C, a_layout=al, b_layout=1, workspace=ws)
        torch.cuda.synchronize()
        res.append(C.clone())
        del keep
    assert torch.equal(res[0], res[1])
    a_ref = a0.float() if form == "dgrad" else (a0.float().t() if form == "wgrad" else a0[:, M:].float().t())
    ref = a_ref @ b0.float()
    assert float((res[1].float() - ref).abs().max()) <= (2e-2 if form == "dgrad" else 2e-3) * float(ref.abs().max())


@pytest.mark.parametrize("T_len,Smax", [(512, 512), (540, 576), (64, 64), (33, 64)])
def test_decode_attention_cache_at_the_end_of_its_allocation(ops, T_len, Smax):
    """Single-query attention against the K/V cache [B, H, Smax, hd] (decode.attn_decode): cache, query rows, mask and output each end an allocation;
    T_len == Smax is the aligned case (the last key row is the cache's last row)."""
    from egoscaler_amd import decode
    B, H, hd = 4, 4, 128
    d = H * hd
    kc0 = rnd(B, H, Smax, hd, seed=5).cuda()
    vc0 = rnd(B, H, Smax, hd, seed=6).cuda()
    qkv0 = rnd(B, 3 * d, seed=7).cuda()
    km0 = torch.ones(B, Smax, dtype=torch.uint8, device="cuda")
    km0[1, 2:5] = 0
    res = []
    for placed in (False, True):
        keep = []
        put = (lambda t: at_end(t, keep)) if placed else (lambda t: t.clone())
        kc, vc, qkv, km = put(kc0), put(vc0), put(qkv0), put(km0)
        out = put(torch.full((B, d), 3.0, dtype=torch.bfloat16, device="cuda"))
        decode.attn_decode(qkv, 3 * d, kc, vc, km, out, B, H, hd, Smax, T_len, hd ** -0.5)
        torch.cuda.synchronize()
        res.append(out.clone())
        del keep
    assert torch.equal(res[0], res[1])
    q = qkv0[:, :d].float().view(B, H, 1, hd)
    sc = (q @ kc0.float()[:, :, :T_len].transpose(-1, -2)) * hd ** -0.5
    sc = sc.masked_fill(~km0[:, None, None, :T_len].bool(), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vc0.float()[:, :, :T_len]).reshape(B, d)
    assert float((res[1].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
