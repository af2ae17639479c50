"""GPU: every dense / row kernel of libegomi.so against a plain PyTorch fp32 CPU evaluation of the
same op.  fp32 kernels: <= 1e-4 relative to the output scale (target of the path: 1e-3);
bf16 kernels: inputs are bf16-rounded first, tolerance 2e-2 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from egoscaler_amd import ops as O
    return O


def close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 96), (77, 53, 40), (513, 1152, 384), (1, 8, 8), (300, 32262 // 16, 128)])
@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(ops, dtype, M, N, K, al, bl):
    a = rnd(M, K, dtype=dtype, seed=1)
    b = rnd(N, K, dtype=dtype, seed=2)
    ref = a.float() @ b.float().t()
    A = (a if al == 0 else a.t().contiguous()).cuda()
    B = (b if bl == 0 else b.t().contiguous()).cuda()
    out = ops.mm(A, B, a_layout=al, b_layout=bl)
    close(out, ref, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues_and_views(ops, dtype):
    M, N, K = 150, 200, 72
    a, w, bias, res = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=3), rnd(N, dtype=dtype, seed=4), rnd(M, N, dtype=dtype, seed=5)
    A, W, Bi, R = a.cuda(), w.cuda(), bias.cuda(), res.cuda()
    lin = a.float() @ w.float().t()
    close(ops.mm(A, W, bias=Bi), lin + bias.float(), TOL[dtype])
    close(ops.mm(A, W, bias=Bi, act=ops.ACT_GELU), F.gelu(lin + bias.float()), TOL[dtype])
    close(ops.mm(A, W, bias=Bi, act=ops.ACT_RELU), F.relu(lin + bias.float()), TOL[dtype])
    close(ops.mm(A, W, bias=Bi, residual=R, alpha=0.5), 0.5 * lin + bias.float() + res.float(), TOL[dtype])
    # output into a column slice of a wider buffer (ldc > N), fp32 accumulate on top of existing C
    big = torch.zeros(M, 3 * N, dtype=dtype, device="cuda")
    ops.mm(A, W, out=big[:, N:2 * N])
    close(big[:, N:2 * N], lin, TOL[dtype])
    assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0
    acc = torch.ones(M, N, dtype=torch.float32, device="cuda")
    ops.mm(A, W, out=acc, accumulate=True)
    close(acc, lin + 1.0, TOL[dtype])
    # unaligned operand (odd leading dimension) goes through the guarded path
    wide = torch.zeros(M, K + 3, dtype=dtype, device="cuda")
    wide[:, 1:K + 1] = A
    close(ops.mm(wide[:, 1:K + 1], W), lin, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_batched_attention_pattern(ops, dtype):
    Bq, H, Sq, hd = 2, 3, 70, 32
    d = H * hd
    qkv = rnd(Bq, Sq, 3 * d, dtype=dtype, seed=7)
    Q = qkv.cuda()
    sc = torch.empty(Bq * H, Sq, Sq, dtype=torch.float32, device="cuda") if dtype == torch.bfloat16 else torch.empty(Bq * H, Sq, Sq, device="cuda")
    q, k, v = Q[:, :, :d], Q[:, :, d:2 * d], Q[:, :, 2 * d:]
    ops.gemm_raw(q, k, sc, Sq, Sq, hd, 3 * d, 3 * d, Sq, 0, 0, alpha=0.25, batch=Bq * H, batch_inner=H,
                 strides=(Sq * 3 * d, hd, Sq * 3 * d, hd, H * Sq * Sq, Sq * Sq))
    qf = qkv.float().view(Bq, Sq, 3, H, hd)
    ref = torch.einsum("bqhd,bkhd->bhqk", qf[:, :, 0], qf[:, :, 1]) * 0.25
    close(sc.view(Bq, H, Sq, Sq), ref, TOL[dtype])
    # P.V with V as the [K,N] operand, output written head-interleaved [B,S,H,hd]
    Pm = torch.softmax(ref, -1).to(dtype)
    out = torch.empty(Bq, Sq, d, dtype=dtype, device="cuda")
    ops.gemm_raw(Pm.cuda(), v, out, Sq, hd, Sq, Sq, 3 * d, d, 0, 1, batch=Bq * H, batch_inner=H,
                 strides=(H * Sq * Sq, Sq * Sq, Sq * 3 * d, hd, Sq * d, hd))
    refo = torch.einsum("bhqk,bkhd->bqhd", Pm.float(), qf[:, :, 2]).reshape(Bq, Sq, d)
    close(out, refo, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_and_rmsnorm(ops, dtype):
    x, add, w, b = rnd(37, 384, dtype=dtype), rnd(37, 384, dtype=dtype, seed=1), rnd(384, dtype=dtype, seed=2), rnd(384, dtype=dtype, seed=3)
    so = torch.empty(37, 384, dtype=dtype, device="cuda")
    y = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5, add=add.cuda(), sum_out=so)
    s = (x.float() + add.float()).to(dtype).float()
    close(so, s, TOL[dtype])
    close(y, F.layer_norm(s, (384,), w.float(), b.float(), 1e-5), TOL[dtype])
    close(ops.layernorm(x.cuda(), w.cuda(), b.cuda()), F.layer_norm(x.float(), (384,), w.float(), b.float(), 1e-5), TOL[dtype])
    # RMSNorm fwd/bwd vs autograd
    R, C = 45, 512
    xr, wr, dy, dadd = rnd(R, C, dtype=dtype, seed=4), (1 + 0.1 * rnd(C, seed=5)).to(dtype), rnd(R, C, dtype=dtype, seed=6), rnd(R, C, dtype=dtype, seed=7)
    xf, wf = xr.float().requires_grad_(True), wr.float().requires_grad_(True)
    yf = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    yf.backward(dy.float())
    rstd = torch.empty(R, dtype=torch.float32, device="cuda")
    close(ops.rmsnorm(xr.cuda(), wr.cuda(), 1e-6, rstd=rstd), yf.detach(), TOL[dtype])
    dw = torch.zeros(C, dtype=torch.float32, device="cuda")
    dx = ops.rmsnorm_bwd(dy.cuda(), xr.cuda(), wr.cuda(), rstd, dx_add=dadd.cuda(), dw=dw)
    close(dx, xf.grad + dadd.float(), TOL[dtype])
    close(dw, wf.grad, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_matches_oracle_and_inverse(ops, dtype):
    from oracle import llama as OL
    B, S, H, hd = 2, 19, 4, 32
    q = rnd(B, S, H, hd, dtype=dtype)
    cos, sin = OL.rope_cos_sin(S, hd, 10000.0, offset=3, dtype=dtype)
    qo, _ = OL.apply_rope(q.transpose(1, 2), q.transpose(1, 2), cos, sin)
    ct, st = ops.rope_tables(64, hd, 10000.0)
    buf = torch.zeros(B * S, 3 * H * hd, dtype=dtype, device="cuda")
    buf[:, H * hd:2 * H * hd] = q.reshape(B * S, H * hd).cuda()
    view = buf[:, H * hd:2 * H * hd]
    ops.rope_(view, ct.cuda(), st.cuda(), B * S, S, 3, H, hd, 3 * H * hd)
    got = view.reshape(B, S, H, hd)
    if dtype == torch.float32:
        close(got, qo.transpose(1, 2), 1e-6)
    else:
        assert torch.equal(got.cpu(), qo.transpose(1, 2).contiguous()), "bf16 RoPE must reproduce HF's rounding sequence"
    ops.rope_(view, ct.cuda(), st.cuda(), B * S, S, 3, H, hd, 3 * H * hd, inverse=True)
    close(view.reshape(B, S, H, hd), q, 2e-2 if dtype == torch.bfloat16 else 1e-5)
    assert float(buf[:, :H * hd].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_gelu(ops, dtype):
    R, Fd = 33, 352
    gu, dact = rnd(R, 2 * Fd, dtype=dtype), rnd(R, Fd, dtype=dtype, seed=2)
    g, u = gu[:, :Fd].float().requires_grad_(True), gu[:, Fd:].float().requires_grad_(True)
    ref = F.silu(g) * u
    ref.backward(dact.float())
    GU = gu.cuda()
    out = torch.empty(R, Fd, dtype=dtype, device="cuda")
    ops.swiglu(GU[:, :Fd], GU[:, Fd:], out)
    close(out, ref.detach(), TOL[dtype])
    dgu = torch.empty(R, 2 * Fd, dtype=dtype, device="cuda")
    ops.swiglu_bwd(dact.cuda(), GU[:, :Fd], GU[:, Fd:], dgu[:, :Fd], dgu[:, Fd:])
    close(dgu[:, :Fd], g.grad, TOL[dtype])
    close(dgu[:, Fd:], u.grad, TOL[dtype])
    x = rnd(1000, dtype=dtype, seed=3, scale=2.0)
    xf = x.float().requires_grad_(True)
    y = F.gelu(xf)
    y.backward(torch.ones_like(y) * 0.7)
    close(ops.gelu(x.cuda()), y.detach(), TOL[dtype])
    close(ops.gelu_bwd(torch.full_like(x, 0.7).cuda(), x.cuda()), xf.grad, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_masks_and_backward(ops, dtype):
    B, H, Sq, Sk = 2, 3, 17, 17
    sc = rnd(B * H, Sq, Sk, seed=1, scale=3.0)
    km = torch.ones(B, Sk, dtype=torch.uint8)
    km[1, 12:] = 0
    keep = torch.tril(torch.ones(Sq, Sk, dtype=torch.bool))[None, None] & km.bool()[:, None, None, :]
    ref = torch.softmax(sc.view(B, H, Sq, Sk).masked_fill(~keep, float("-inf")), -1)
    out = torch.empty(B * H, Sq, Sk, dtype=dtype, device="cuda")
    ops.softmax(sc.cuda(), B * H, H, Sq, Sk, out, causal=True, key_mask=km.cuda())
    close(out.view(B, H, Sq, Sk), ref, TOL[dtype])
    out2 = torch.empty(B * H, Sq, Sk, dtype=dtype, device="cuda")
    ops.softmax(sc.cuda(), B * H, H, Sq, Sk, out2)
    close(out2, torch.softmax(sc, -1), TOL[dtype])
    # decode row: one query at offset 9 sees keys 0..9
    o3 = torch.empty(B * H, 1, Sk, dtype=dtype, device="cuda")
    ops.softmax(sc[:, :1].contiguous().cuda(), B * H, H, 1, Sk, o3, causal=True, q_offset=9)
    r3 = torch.softmax(sc[:, :1].masked_fill(torch.arange(Sk)[None, None] > 9, float("-inf")), -1)
    close(o3, r3, TOL[dtype])
    dP = rnd(B * H, Sq, Sk, seed=2)
    Pm = torch.softmax(sc, -1).to(dtype)
    dS = torch.empty_like(out2)
    ops.softmax_bwd(Pm.cuda(), dP.cuda(), dS, B * H * Sq, Sk)
    pf = Pm.float()
    close(dS, pf * (dP - (pf * dP).sum(-1, keepdim=True)), TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_splice_forward_backward_and_errors(ops, dtype):
    from egoscaler_amd.config import dims_tiny
    from egoscaler_amd import synth
    from oracle import pointllm as OPL
    dims = dims_tiny()
    tok, Pn, d, V = dims.tok, dims.pb.point_token_len, 64, dims.lm.vocab_size
    ids, _, _ = synth.synth_batch(dims, 3, text_len=8, num_steps=4, max_traj_token=40)
    ids[2, (ids[2] >= tok.point_patch) & (ids[2] <= tok.point_end)] = 9          # text-only sample
    W, feats = rnd(V, d, dtype=dtype), rnd(3, Pn, d, dtype=dtype, seed=1)
    sp, err, cloud = ops.splice_scan(ids.cuda(), tok, Pn)
    ref_pos = OPL.splice_positions(ids, tok, Pn)
    assert err.cpu().tolist() == [0, 0, 0] and sp.cpu().tolist() == [p[0] if p else -1 for p in ref_pos] and cloud.cpu().tolist() == [0, 1, 2]
    out = ops.embed_splice(ids.cuda(), W.cuda(), feats.cuda(), sp, Pn, cloud_idx=cloud)
    ref = OPL.splice(ids, F.embedding(ids, W), feats, tok, Pn)
    assert torch.equal(out.cpu(), ref), "splice is a pure copy: must be bit-exact"
    dout = rnd(3, ids.shape[1], d, dtype=dtype, seed=2)
    Wf, ff = W.float().requires_grad_(True), feats.float().requires_grad_(True)
    OPL.splice(ids, F.embedding(ids, Wf), ff, tok, Pn).backward(dout.float())
    dW = torch.zeros(V, d, dtype=torch.float32, device="cuda")
    df = torch.zeros(3, Pn, d, dtype=dtype, device="cuda")
    ops.embed_splice_bwd(dout.cuda(), ids.cuda(), sp, Pn, V, dW, df, cloud_idx=cloud)
    close(dW, Wf.grad, 1e-5)
    close(df, ff.grad, 1e-6)
    bad = ids.clone()
    bad[0, (bad[0] == tok.point_end).nonzero()[0, 0]] = 5            # count mismatch (pointllm.py:146)
    bad[1, (bad[1] == tok.point_end).nonzero()[0, 0]] = 5
    bad[1, -1] = tok.point_end                                       # end token in the wrong place (:150)
    _, err, _ = ops.splice_scan(bad.cuda(), tok, Pn)
    assert err.cpu().tolist() == [1, 2, 0]
    # several segments in a sample (the reference's running cloud index, pointllm.py:135-156): sample 0 = [seg, seg], sample 1 text only,
    # sample 2 = [seg] -> last segment of sample 0 gets cloud 0, sample 2 gets cloud 3; with 3 clouds sample 2 is an IndexError (code 4)
    seg = [tok.point_start] + [tok.point_patch] * Pn + [tok.point_end]
    rows = [[1, 7] + seg + [8, 9] + seg + [10], [1] + [11] * 30, [1, 12, 13] + seg + [14]]
    Sm = max(len(r) for r in rows)
    mids = torch.zeros(3, Sm, dtype=torch.long)
    for i, r in enumerate(rows):
        mids[i, :len(r)] = torch.tensor(r)
    sp, err, cloud = ops.splice_scan(mids.cuda(), tok, Pn, 4)
    assert err.cpu().tolist() == [0, 0, 0] and sp.cpu().tolist() == [2 + len(seg) + 2, -1, 3] and cloud.cpu().tolist() == [0, 2, 3]
    feats4 = rnd(4, Pn, d, dtype=dtype, seed=4)
    out = ops.embed_splice(mids.cuda(), W.cuda(), feats4.cuda(), sp, Pn, cloud_idx=cloud)
    assert torch.equal(out.cpu(), OPL.splice(mids, F.embedding(mids, W), feats4, tok, Pn))
    sp, err, cloud = ops.splice_scan(mids.cuda(), tok, Pn, 3)
    assert err.cpu().tolist() == [0, 0, 4] and sp.cpu().tolist()[2] == -1
    with pytest.raises(IndexError):
        OPL.splice(mids, F.embedding(mids, W), feats4[:3], tok, Pn)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_entropy_ignore_index(ops, dtype):
    R, V = 50, 1000
    lg = rnd(R, V, dtype=dtype, scale=2.0)
    tg = torch.randint(1, V, (R,), generator=torch.Generator().manual_seed(1))
    tg[::7] = 0
    lf = lg.float().requires_grad_(True)
    loss = F.cross_entropy(lf, tg, ignore_index=0)
    loss.backward()
    dl = torch.empty(R, V, dtype=dtype, device="cuda")
    ls, cnt = ops.cross_entropy(lg.cuda(), tg.cuda(), 0, dl)
    assert int(cnt) == int((tg != 0).sum())
    assert abs(float(ls) / int(cnt) - float(loss)) < 1e-4 * max(1, abs(float(loss)))
    close(dl, lf.grad, TOL[dtype])


def test_adamw_matches_torch(ops):
    n = 5000
    p0, g = rnd(n), rnd(n, seed=1)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    master, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    copy = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        p.grad = g.clone() * step
        opt.step()
        ops.adamw(master, copy, (g * step).cuda(), m, v, 2e-3, 0.9, 0.999, 1e-8, 0.01, step)
    close(master, p.detach(), 1e-5)
    assert torch.equal(copy.cpu(), master.cpu().bfloat16())


@pytest.mark.parametrize("n,off,with_copy", [(5003, 0, True), (4096, 0, False), (1001, 1, True), (3, 0, True), (70000, 4, True)])
def test_adamw_vector_form_equals_scalar_form(ops, n, off, with_copy):
    """The 16-B form (aligned buffers, n >= 4; a scalar launch finishes n % 4) against the 4-B form, which the same call takes when a
    buffer is not 16-B aligned (views at an odd element offset): identical bits in master, m, v and the bf16 model copy."""
    def at(vals, o, dtype=torch.float32):                  # the same values, placed o elements into a fresh buffer
        buf = torch.zeros(n + 8, dtype=dtype, device="cuda")
        buf[o:o + n] = vals.cuda().to(dtype)
        return buf[o:o + n]
    vals = [rnd(n, seed=s) for s in (0, 1, 2, 3)]
    outs = []
    for o in (off, 1 if off != 1 else 3):                  # second run: forced onto the scalar form by an odd offset
        master, m, v, g = at(vals[0], o), at(vals[1], o), at(vals[2].abs(), o), at(vals[3], o)
        cp = at(torch.zeros(n), o, torch.bfloat16)
        for step in (1, 2):
            ops.adamw(master, cp if with_copy else None, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, 0.5)
        outs.append([t.clone() for t in (master, m, v, cp)])
    if off == 1:
        return                                              # both runs scalar: nothing to compare beyond not faulting
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_cast_add_groupmax_smallk(ops, dtype):
    x = rnd(70, 45, dtype=dtype)
    t = ops.transpose(x.cuda(), ldo=96)
    assert torch.equal(t[:, :70].cpu(), x.t()) and float(t[:, 70:].abs().max()) == 0
    assert torch.equal(ops.cast(x.cuda(), torch.float32).cpu(), x.float())
    assert torch.equal(ops.cast(x.float().cuda(), torch.bfloat16).cpu(), x.float().bfloat16())
    close(ops.add(x.cuda(), x.cuda()), 2 * x.float(), TOL[dtype])
    BG, M, C = 6, 16, 40
    h = rnd(BG * M, C, dtype=dtype, seed=2)
    hv = h.view(BG, M, C)
    assert torch.equal(ops.group_max(h.cuda(), BG, M, C).cpu(), hv.max(1)[0])
    cat = torch.cat([hv.max(1, keepdim=True)[0].expand(-1, M, -1), hv], -1).reshape(BG * M, 2 * C)
    assert torch.equal(ops.group_max(h.cuda(), BG, M, C, concat=True).cpu(), cat)
    xs, w, b = rnd(100, 3), rnd(128, 3, dtype=dtype, seed=3), rnd(128, dtype=dtype, seed=4)
    close(ops.linear_smallk(xs.cuda(), w.cuda(), b.cuda(), act=ops.ACT_GELU), F.gelu(xs @ w.float().t() + b.float()), TOL[dtype])
    x6 = rnd(64, 6, dtype=dtype, seed=5)
    w6 = rnd(128, 6, dtype=dtype, seed=6)
    close(ops.linear_smallk(x6.cuda(), w6.cuda(), b.cuda(), act=ops.ACT_RELU), F.relu(x6.float() @ w6.float().t() + b.float()), TOL[dtype])


@pytest.mark.parametrize("M,N,K", [(700, 4096, 4096), (128, 128, 64), (1280, 1024, 11008), (300, 2018, 384), (5536, 256, 512)])
def test_gemm_tuned_nt_kernel(ops, M, N, K):
    """The tuned bf16 NT kernel (LDS-DMA staging, swizzled LDS, clamped edge rows) against fp32 CPU
    matmul and against the generic kernel on the same inputs."""
    a, w = rnd(M, K, dtype=torch.bfloat16, seed=11), rnd(N, K, dtype=torch.bfloat16, seed=12, scale=0.05)
    bias, res = rnd(N, dtype=torch.bfloat16, seed=13), rnd(M, N, dtype=torch.bfloat16, seed=14)
    A, W, Bi, R = a.cuda(), w.cuda(), bias.cuda(), res.cuda()
    ref = a.float() @ w.float().t()
    fast = ops.mm(A, W)
    gen = ops.mm(A, W, force_generic=True)
    close(fast, ref, 2e-2)
    assert float((fast.float() - gen.float()).abs().max()) <= 2e-2 * float(ref.abs().max())
    close(ops.mm(A, W, bias=Bi, act=ops.ACT_GELU, residual=R, alpha=0.5), torch.nn.functional.gelu(0.5 * ref + bias.float()) + res.float(), 2e-2)
    acc = torch.full((M, N), 2.0, dtype=torch.float32, device="cuda")
    ops.mm(A, W, out=acc, accumulate=True)
    close(acc, ref + 2.0, 2e-2)
    big = torch.zeros(M, N + 64, dtype=torch.bfloat16, device="cuda")
    ops.mm(A, W, out=big[:, 32:32 + N] if N % 4 == 0 else big[:, :N])
    close(big[:, 32:32 + N] if N % 4 == 0 else big[:, :N], ref, 2e-2)
    # kernel selection is what the test thinks it is
    d = ops.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), fast.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
    d.ab_dtype, d.c_dtype, d.batch = 1, 1, 1
    import ctypes
    from egoscaler_amd import _lib
    assert _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d)) == 1


def test_fused_attention_backward_with_inverse_rope(ops):
    """egomi_attn_bwd with rope tables: dq and dk equal the plain kernel's output followed by egomi_rope(inverse=1), bit for bit;
    dv is untouched."""
    B, S, H, hd = 2, 200, 3, 128
    d = H * hd
    qkv = rnd(B * S, 3 * d, dtype=torch.bfloat16, seed=5).cuda()
    dout = rnd(B * S, d, dtype=torch.bfloat16, seed=6, scale=0.1).cuda()
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=True)
    delta = torch.empty_like(lse)
    cos, sin = ops.rope_tables(256, hd, 10000.0)
    cos, sin = cos.cuda(), sin.cuda()
    plain = torch.zeros_like(qkv)
    ops.attn_bwd(qkv, out, lse, dout, plain, delta, B, S, H, hd, hd ** -0.5, causal=True)
    ops.rope_(plain, cos, sin, B * S, S, 0, 2 * H, hd, 3 * d, inverse=True)
    fused = torch.zeros_like(qkv)
    ops.attn_bwd(qkv, out, lse, dout, fused, delta, B, S, H, hd, hd ** -0.5, causal=True, rope=(cos, sin))
    assert torch.equal(fused, plain)
    with pytest.raises(ValueError):
        ops.attn_bwd(qkv, out, lse, dout, fused, delta, B, S, H, hd, hd ** -0.5, causal=True, rope=(cos[:100], sin[:100]))


def _kernel_id(ops, A, W, C, M, N, K):
    import ctypes
    from egoscaler_amd import _lib
    d = ops.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, C.stride(0)
    d.ab_dtype, d.c_dtype, d.batch = 1, 1 if C.dtype == torch.bfloat16 else 0, 1
    return _lib.lib().egomi_gemm_kernel_id(ctypes.byref(d))


@pytest.mark.parametrize("M,N,K", [(2900, 4100, 2048), (3000, 3000, 2112), (5536, 4096, 4096)])
def test_gemm_8phase_kernel(ops, M, N, K):
    """256x256 8-phase kernel (egomi_gemm_kernel_id == 2): ragged M and N, odd K-tile count (2112 = 33 x 64), every
    epilogue form, fp32 output, and the K-sliced tail rows (library plan, explicit plans rows*16+S, no workspace)."""
    a, w = rnd(M, K, dtype=torch.bfloat16, seed=21), rnd(N, K, dtype=torch.bfloat16, seed=22, scale=0.05)
    bias, res = rnd(N, dtype=torch.bfloat16, seed=23), rnd(M, N, dtype=torch.bfloat16, seed=24)
    A, W, Bi, R = a.cuda(), w.cuda(), bias.cuda(), res.cuda()
    ref = (A.float() @ W.float().t()).cpu()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    assert _kernel_id(ops, A, W, out, M, N, K) == 2
    close(ops.mm(A, W, out=out), ref, 2e-2)
    gen = ops.mm(A, W, force_generic=True)
    assert float((out.float() - gen.float()).abs().max()) <= 2e-2 * float(ref.abs().max())
    close(ops.mm(A, W, bias=Bi, act=ops.ACT_GELU, residual=R, alpha=0.5), F.gelu(0.5 * ref + bias.float()) + res.float(), 2e-2)
    close(ops.mm(A, W, residual=R), ref + res.float(), 2e-2)                       # interior fast-path epilogue, residual
    acc = R.clone()
    close(ops.mm(A, W, out=acc, accumulate=True), ref + res.float(), 2e-2)          # ... and accumulate
    f32 = torch.full((M, N), 2.0, dtype=torch.float32, device="cuda")
    ops.mm(A, W, out=f32, accumulate=True)
    close(f32, ref + 2.0, 1e-3)
    ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    tm = (M + 255) // 256
    for rows, S in ((1, 2), (2, 4), (min(tm, 5), 3), (tm, 2)):                        # explicit tail plans, up to every row sliced
        got = ops.mm(A, W, bias=Bi, residual=R, workspace=ws, split_k=rows * 16 + S)
        close(got, ref + bias.float() + res.float(), 2e-2)
    # deterministic: same plan, same bits
    x1 = ops.mm(A, W, workspace=ws, split_k=2 * 16 + 2).clone()
    x2 = ops.mm(A, W, workspace=ws, split_k=2 * 16 + 2)
    assert torch.equal(x1, x2)


def test_gemm_time_next_brackets_the_256_kernel_alone(ops):
    """egomi_gemm_time_next (include/egomi.h): the library records the two events around the 256x256 kernel itself — shorter than
    the call when the tail rows are K-sliced (separate combine pass), one-shot, untouched by other kernels."""
    torch.manual_seed(0)
    A = torch.randn(5536, 4096, device="cuda").bfloat16()
    W = (torch.randn(4096, 4096, device="cuda") * 0.02).bfloat16()
    prof = ops.GemmProfiler(min_flops=0, kernel_ids=(1, 2))
    ops.PROFILER = prof
    try:
        for _ in range(3):
            ops.mm(A, W)                                                   # kernel id 2 (+ combine of the sliced tail rows)
        ops.mm(A[:256], W)                                                 # kernel id 1: no library events, the call is the bracket
    finally:
        ops.PROFILER = None
    assert [r[3] is not None for r in prof.recs] == [True, True, True, False]
    sm = prof.summary()
    assert sm["launches"] == 4 and 0 < sm["ms"] < sm["call_ms"]
    # one-shot: a request made before a call that takes another kernel is consumed by that call and never fires later
    import ctypes
    from egoscaler_amd import _lib
    L = _lib.lib()
    k0, k1 = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.egomi_event_create(ctypes.byref(k0)) == 0 and L.egomi_event_create(ctypes.byref(k1)) == 0
    assert L.egomi_gemm_time_next(k0, k1) == 0
    ops.mm(A[:256], W)
    ops.mm(A, W)
    torch.cuda.synchronize()
    t = ctypes.c_float()
    assert L.egomi_event_elapsed_ms(k0, k1, ctypes.byref(t)) != 0          # never recorded
    assert L.egomi_gemm_time_next(k0, None) != 0                           # both or neither
    L.egomi_event_destroy(k0); L.egomi_event_destroy(k1)


@pytest.mark.parametrize("M,N,K", [(5536, 4096, 4096), (5536, 12288, 4096), (2900, 4100, 2048), (4096, 4096, 2048), (1500, 8200, 4224),
                                   (5536, 4096, 11008), (3000, 3000, 2112)])
def test_gemm_persistent_8phase_kernel(ops, M, N, K):
    """Persistent form of the 256x256 kernel (one block per CU walking whole tiles + an in-launch, last-arriver reduction of
    the K-sliced remainder): one full round + remainder, several rounds, fewer tiles than CUs (pure stream-K), an exact
    multiple of the CU count (no remainder), ragged M / N edges, long K, and an odd K-tile count (falls back).  Every
    epilogue form; repeated launches are bit-identical (fixed summation order) and leave the ticket words zero."""
    a, w = rnd(M, K, dtype=torch.bfloat16, seed=31), rnd(N, K, dtype=torch.bfloat16, seed=32, scale=0.05)
    bias, res = rnd(N, dtype=torch.bfloat16, seed=33), rnd(M, N, dtype=torch.bfloat16, seed=34)
    A, W, Bi, R = a.cuda(), w.cuda(), bias.cuda(), res.cuda()
    ref = (A.float() @ W.float().t()).cpu()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    assert _kernel_id(ops, A, W, out, M, N, K) == 2
    close(ops.mm(A, W, out=out, persistent=True), ref, 2e-2)
    old = ops.mm(A, W, persistent=False)                                            # non-persistent kernel, same operands
    assert float((out.float() - old.float()).abs().max()) <= 8e-3 * float(ref.abs().max())
    close(ops.mm(A, W, bias=Bi, act=ops.ACT_GELU, residual=R, alpha=0.5, persistent=True), F.gelu(0.5 * ref + bias.float()) + res.float(), 2e-2)
    close(ops.mm(A, W, residual=R, persistent=True), ref + res.float(), 2e-2)
    acc = R.clone()
    close(ops.mm(A, W, out=acc, accumulate=True, persistent=True), ref + res.float(), 2e-2)
    f32 = torch.full((M, N), 2.0, dtype=torch.float32, device="cuda")
    ops.mm(A, W, out=f32, accumulate=True, persistent=True)
    close(f32, ref + 2.0, 1e-3)
    big = torch.zeros(M, N + 64, dtype=torch.bfloat16, device="cuda")               # output into a column window (ldc > N)
    ops.mm(A, W, out=big[:, 32:32 + N], persistent=True)
    close(big[:, 32:32 + N], ref, 2e-2)
    assert float(big[:, :32].abs().max()) == 0 and float(big[:, 32 + N:].abs().max()) == 0
    x1 = ops.mm(A, W, out_dtype=torch.float32, persistent=True).clone()
    for _ in range(5):
        assert torch.equal(ops.mm(A, W, out_dtype=torch.float32, persistent=True), x1)              # fp32 output: any change of summation order would show
    close(ops.mm(A, W), ref, 2e-2)                                                   # whatever the library's own rule picks
    torch.cuda.synchronize()
    assert int(ops._tail_workspace(A.device)[:1024].view(torch.int32).abs().max()) == 0


def _interleave32(gate, up):
    Fd = gate.shape[-1]
    return torch.stack([gate.reshape(*gate.shape[:-1], Fd // 32, 32), up.reshape(*up.shape[:-1], Fd // 32, 32)], -2).reshape(*gate.shape[:-1], 2 * Fd)


def test_swiglu_interleaved32_layout(ops):
    """egomi_swiglu_il_fwd / _bwd on the interleaved-32 gate|up layout == the plain kernels on the de-interleaved halves, bit for bit."""
    M, Fd = 300, 352
    gate, up, dact = rnd(M, Fd, dtype=torch.bfloat16, seed=41).cuda(), rnd(M, Fd, dtype=torch.bfloat16, seed=42).cuda(), rnd(M, Fd, dtype=torch.bfloat16, seed=43).cuda()
    gu = _interleave32(gate, up).contiguous()
    ref = torch.empty(M, Fd, dtype=torch.bfloat16, device="cuda")
    ops.swiglu(gate, up, ref)
    got = ops.swiglu_il(gu, torch.empty_like(ref))
    assert torch.equal(got, ref)
    close(got, F.silu(gate.float().cpu()) * up.float().cpu(), 1e-2)
    dg, du = torch.empty_like(gate), torch.empty_like(up)
    ops.swiglu_bwd(dact, gate, up, dg, du)
    dgu = ops.swiglu_il_bwd(dact, gu, torch.empty_like(gu))
    assert torch.equal(dgu, _interleave32(dg, du))


@pytest.mark.parametrize("tall_mode", [0, 1, 2])
@pytest.mark.parametrize("M,Fd,K", [(5536, 11008, 4096), (5000, 2048, 2048), (2304, 8192, 2112)])
def test_gemm_swiglu_epilogue(ops, M, Fd, K, tall_mode, request):
    """EGOMI_EPI_SWIGLU: x . [Wgate;Wup]^T with the stacked rows interleaved in blocks of 32 — C (gate|up, interleaved-32) equals
    the plain product and C2 equals egomi_swiglu_il_fwd(C) bit for bit, on whole tiles, ragged M and K-sliced tail rows; shapes
    the fused path cannot serve are refused, never silently computed without C2."""
    from egoscaler_amd import _lib
    import ctypes
    x = rnd(M, K, dtype=torch.bfloat16, seed=51).cuda()
    wg, wu = rnd(Fd, K, dtype=torch.bfloat16, seed=52, scale=0.05).cuda(), rnd(Fd, K, dtype=torch.bfloat16, seed=53, scale=0.05).cuda()
    w = torch.stack([wg.view(Fd // 32, 32, K), wu.view(Fd // 32, 32, K)], 1).reshape(2 * Fd, K).contiguous()
    assert ops.gemm_kernel_id(M, 2 * Fd, K) == 2
    # (the plain product and the fused one take the same form and the same K-sliced rows in every mode: 0 = 256x256 tiles only, 1 = the library's
    #  choice — 352x256 tiles, or both forms side by side on a column split —, 2 = 352x256 tiles wherever they apply)
    _lib.lib().egomi_gemm_set_tall(ctypes.c_int(tall_mode))
    request.addfinalizer(lambda: _lib.lib().egomi_gemm_set_tall(ctypes.c_int(-1)))
    gu_ref = ops.mm(x, w)
    act_ref = ops.swiglu_il(gu_ref, torch.empty(M, Fd, dtype=torch.bfloat16, device="cuda"))
    gu = torch.empty_like(gu_ref)
    act = torch.full((M, Fd), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.mm(x, w, out=gu, swiglu_out=act)
    assert torch.equal(gu, gu_ref)
    assert torch.equal(act, act_ref)
    g32, u32 = x.float() @ wg.float().t(), x.float() @ wu.float().t()
    close(act, (F.silu(g32) * u32).cpu(), 2e-2)                                    # and it is the right function of the operands
    with pytest.raises(_lib.EgomiError):
        ops.mm(x[:64], w, out=gu[:64], swiglu_out=act[:64])                          # too small for the 256x256 kernel
    with pytest.raises(_lib.EgomiError):
        ops.mm(x, w, out=gu, swiglu_out=act, bias=torch.zeros(2 * Fd, dtype=torch.bfloat16, device="cuda"))


@pytest.mark.parametrize("tall_mode", [0, 1, 2])
@pytest.mark.parametrize("M,Fd,K", [(5536, 11008, 4096), (4500, 2880, 2048), (2304, 4096, 2112), (4100, 4160, 2048), (5536, 4096, 2048)])
def test_gemm_swiglu_backward_epilogue(ops, M, Fd, K, tall_mode, request):
    """EGOMI_EPI_SWIGLU_BWD: the down_proj data gradient dx . W_down with SwiGLU's backward in its epilogue — d(gate|up) equals
    egomi_swiglu_il_bwd applied to the stored bf16 product, bit for bit (whole tiles, ragged M, a last column tile that is partly outside
    (Fd % 256 != 0), K-sliced tail rows), d(act) is never written, and it is the right function of the operands; refused where the
    256x256 kernel does not run."""
    from egoscaler_amd import _lib
    import ctypes
    _lib.lib().egomi_gemm_set_tall(ctypes.c_int(tall_mode))             # 0 = 256x256 tiles only, 1 = the library's choice (incl. the column split), 2 = 352x256 wherever possible
    request.addfinalizer(lambda: _lib.lib().egomi_gemm_set_tall(ctypes.c_int(-1)))
    dx = rnd(M, K, dtype=torch.bfloat16, seed=61).cuda()
    wt = rnd(Fd, K, dtype=torch.bfloat16, seed=62, scale=0.05).cuda()                 # W_down^T: [ffn, d]
    gate, up = rnd(M, Fd, dtype=torch.bfloat16, seed=63), rnd(M, Fd, dtype=torch.bfloat16, seed=64)
    gu = torch.stack([gate.view(M, Fd // 32, 32), up.view(M, Fd // 32, 32)], 2).reshape(M, 2 * Fd).contiguous().cuda()
    assert ops.gemm_kernel_id(M, Fd, K) == 2
    dact = ops.mm(dx, wt)
    ref = ops.swiglu_il_bwd(dact, gu, torch.empty_like(gu))
    dgu = torch.full((M, 2 * Fd), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.mm(dx, wt, out=dgu, swiglu_bwd_gu=gu)
    assert torch.equal(dgu, ref)
    d32 = dx.float() @ wt.float().t()
    g32, u32 = gate.cuda().float(), up.cuda().float()
    sg = torch.sigmoid(g32)
    want_g, want_u = d32 * u32 * sg * (1 + g32 * (1 - sg)), d32 * g32 * sg
    got = dgu.view(M, Fd // 32, 2, 32).float()
    close(got[:, :, 0].reshape(M, Fd), want_g.cpu(), 3e-2)
    close(got[:, :, 1].reshape(M, Fd), want_u.cpu(), 3e-2)
    with pytest.raises(_lib.EgomiError):
        ops.mm(dx[:64], wt, out=dgu[:64], swiglu_bwd_gu=gu[:64])                      # too small for the 256x256 kernel
    with pytest.raises(_lib.EgomiError):
        ops.mm(dx, wt, out=dgu, swiglu_bwd_gu=gu, accumulate=True)


def _ref_attention(qkv, B, S, H, hd, scale, causal, km):
    x = qkv.float().view(B, S, 3, H, hd)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    sc = (q @ k.transpose(-1, -2)) * scale
    keep = torch.ones(S, S, dtype=torch.bool)
    if causal:
        keep = torch.tril(keep)
    keep = keep[None, None]
    if km is not None:
        keep = keep & km.bool()[:, None, None, :]
    sc = sc.masked_fill(~keep, float("-inf"))
    lse = torch.logsumexp(sc, -1)
    return (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, H * hd), lse


@pytest.mark.parametrize("B,S,H,causal,masked", [(2, 200, 3, True, True), (1, 692, 2, True, False), (2, 64, 1, False, True), (1, 33, 2, True, False), (1, 129, 1, True, True)])
def test_fused_attention_forward(ops, B, S, H, causal, masked):
    hd = 128
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=S)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8)
        km[-1, S - S // 5:] = 0                       # right padding on the last sample
    ref, lse_ref = _ref_attention(qkv, B, S, H, hd, hd ** -0.5, causal, km)
    out = torch.zeros(B * S, H * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    ops.attn_fwd(qkv.cuda(), B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=None if km is None else km.cuda())
    rows = torch.ones(B, S, dtype=torch.bool)
    if km is not None and not causal:
        pass
    close(out, ref, 2e-2)
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2


@pytest.mark.parametrize("hd", [128, 64])
@pytest.mark.parametrize("B,S,H,causal,mask", [(2, 692, 3, True, "tail"), (1, 692, 2, True, None), (2, 513, 6, False, None), (1, 200, 2, True, "holes"),
                                               (2, 64, 1, False, "tail"), (1, 33, 2, True, None), (1, 1, 1, True, None), (2, 300, 2, False, "holes"), (1, 1000, 1, True, "tail")])
def test_fused_attention_forward_second_form_is_bit_identical(ops, hd, B, S, H, causal, mask):
    """attn_fwd2_kernel (straight-line interior / edge / dead tile loops, asm LDS-DMA, early transposed reads) against attn_fwd_kernel: same
    arithmetic in the same order -> the same bits in O and LSE, for interior tiles, diagonal tiles, ragged last tiles, key-padding at the
    tail and in the middle of the sequence, dead waves, both head dims."""
    import ctypes
    from egoscaler_amd import _lib
    L = _lib.lib()
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=S + hd).cuda()
    km = None
    if mask is not None:
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "tail":
            km[-1, S - max(1, S // 5):] = 0
        else:
            km[0, 2:4] = 0
            km[-1, S // 2] = 0
        km = km.cuda()
    res = {}
    try:
        for form in (1, 2):
            assert L.egomi_attn_set_fwd_form(form) == 0
            out = torch.full((B * S, H * hd), 7.0, dtype=torch.bfloat16, device="cuda")
            lse = torch.full((B, H, S), 7.0, dtype=torch.float32, device="cuda")
            ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=km)
            torch.cuda.synchronize()
            res[form] = (out, lse)
    finally:
        L.egomi_attn_set_fwd_form(4)
    assert torch.equal(res[1][0], res[2][0])
    assert torch.equal(res[1][1], res[2][1])
    assert bool(torch.isfinite(res[2][0].float()).all())


@pytest.mark.parametrize("group", [0, 3])
@pytest.mark.parametrize("B,S,H,causal,mask", [(2, 692, 3, True, "tail"), (1, 692, 2, True, None), (2, 513, 2, False, None), (1, 200, 2, True, "holes"),
                                               (2, 64, 1, False, "tail"), (1, 33, 2, True, None), (1, 1, 1, True, None), (2, 300, 2, False, "holes"),
                                               (1, 1000, 1, True, "tail"), (8, 256, 4, True, "tail"), (1, 128, 1, True, None), (1, 32, 1, True, None),
                                               (1, 31, 1, False, None), (3, 97, 2, True, "holes"), (1, 2048, 2, True, None)])
def test_fused_attention_forward_third_form(ops, B, S, H, causal, mask, group):
    """attn_fwd3_kernel (round 4: 32-key ring, scores of tile t+1 under the exponentials of tile t, LAZY running max, query blocks aligned to the
    end of the sequence — rows q < 0 in the first block, dead waves, diagonal / ragged / padded tiles, both block orders) and attn_fwd4_kernel
    (the default at head_dim 128: the same arithmetic in PERSISTENT blocks that walk the work items as one continuous K/V stream, next item's Q
    and key mask prefetched; S > 1024 falls back to the third form) against fp32 torch and against the second form: not the same bits as the
    second form (P is taken against a stale maximum, up to 2^6), the same tolerance; LSE is exact either way; forms 3 and 4 agree bit for bit."""
    from egoscaler_amd import _lib
    L = _lib.lib()
    hd = 128
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=S + 3)
    km = None
    if mask is not None:
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "tail":
            km[-1, S - max(1, S // 5):] = 0
        else:
            km[0, 2:4] = 0
            km[-1, S // 2] = 0
    ref, lse_ref = _ref_attention(qkv, B, S, H, hd, hd ** -0.5, causal, km)
    kmc = None if km is None else km.cuda()
    res = {}
    try:
        for form in (2, 3, 4):
            assert L.egomi_attn_set_fwd_form(form) == 0 and L.egomi_attn_set_fwd_group(group) == 0
            out = torch.full((B * S, H * hd), 7.0, dtype=torch.bfloat16, device="cuda")
            lse = torch.full((B, H, S), 7.0, dtype=torch.float32, device="cuda")
            ops.attn_fwd(qkv.cuda(), B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=kmc)
            torch.cuda.synchronize()
            res[form] = (out.float().cpu(), lse.cpu())
    finally:
        L.egomi_attn_set_fwd_form(4)
        L.egomi_attn_set_fwd_group(0)
    scale = float(ref.abs().max())
    e2, e3 = float((res[2][0] - ref).abs().max()), float((res[3][0] - ref).abs().max())
    assert e3 <= 2e-2 * scale and e3 <= max(2.0 * e2, 1e-2 * scale), (e2, e3, scale)
    assert float((res[3][1] - lse_ref).abs().max()) < 2e-2 and bool(torch.isfinite(res[3][0]).all())
    assert torch.equal(res[3][0], res[4][0]) and torch.equal(res[3][1], res[4][1])        # the persistent form: the same arithmetic per item


@pytest.mark.parametrize("blocks", [1, 3, 7])
@pytest.mark.parametrize("B,S,H,causal,mask", [(2, 692, 3, True, "tail"), (3, 200, 2, True, "holes"), (2, 300, 2, False, "holes"), (4, 33, 2, True, None),
                                               (2, 64, 3, False, "tail"), (1, 1000, 2, True, "tail")])
def test_fused_attention_forward_persistent_blocks_walk_several_items(ops, B, S, H, causal, mask, blocks):
    """attn_fwd4_kernel with its grid capped to 1 / 3 / 7 blocks: every block walks SEVERAL work items (different ranks, heads and samples, so the
    K/V stream crosses item boundaries, the Q prefetch and the two key-mask buffers are exercised, items of one and two tiles follow long ones)
    — bit-identical to the third form, which runs one block per item."""
    from egoscaler_amd import _lib
    L = _lib.lib()
    hd = 128
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=S + 5).cuda()
    km = None
    if mask is not None:
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "tail":
            km[-1, S - max(1, S // 5):] = 0
            km[0, S - 3:] = 0
        else:
            km[0, 2:4] = 0
            km[-1, S // 2] = 0
        km = km.cuda()
    res = {}
    try:
        for form in (3, 4):
            assert L.egomi_attn_set_fwd_form(form) == 0 and L.egomi_attn_set_fwd_blocks(blocks if form == 4 else 0) == 0
            out = torch.full((B * S, H * hd), 7.0, dtype=torch.bfloat16, device="cuda")
            lse = torch.full((B, H, S), 7.0, dtype=torch.float32, device="cuda")
            ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=km)
            torch.cuda.synchronize()
            res[form] = (out, lse)
    finally:
        L.egomi_attn_set_fwd_form(4)
        L.egomi_attn_set_fwd_blocks(0)
    assert torch.equal(res[3][0], res[4][0]) and torch.equal(res[3][1], res[4][1])
    assert bool(torch.isfinite(res[4][0].float()).all())


def test_fused_attention_forward_third_form_rescales_on_a_late_maximum(ops):
    """The lazy maximum's rare branch (cdna_hip_programming.md rule 26): one key far above every earlier score, placed in a late tile, forces
    the rescale of O and l there; a key just UNDER the threshold (2^6 in the exp2 domain) must not.  Full-tensor fp32 reference."""
    B, S, H, hd = 1, 320, 2, 128
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=77).float() * 0.3
    x = qkv.view(B, S, 3, H, hd)
    x[0, 250, 1, 0] = x[0, 300, 0, 0] * 6.0                             # key 250 of head 0 lines up with query 300: a score far above the rest
    x[0, 200, 1, 1] = x[0, 310, 0, 1] * 1.2                             # head 1: a mild outlier, under the threshold
    qkv = qkv.bfloat16()
    ref, lse_ref = _ref_attention(qkv, B, S, H, hd, hd ** -0.5, True, None)
    out = torch.zeros(B * S, H * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    ops.attn_fwd(qkv.cuda(), B, S, H, hd, hd ** -0.5, out, lse, causal=True, key_mask=None)
    close(out, ref, 2e-2)
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2


@pytest.mark.parametrize("B,S,H,causal,masked", [(2, 513, 6, False, False), (1, 65, 2, False, True), (2, 200, 3, True, True), (1, 1, 1, False, False)])
def test_fused_attention_forward_hd64(ops, B, S, H, causal, masked):
    """head_dim 64 instance of the forward kernel (PointBERT blocks, point_encoder.py:36-57: 513 tokens, 6 heads, no mask)."""
    hd = 64
    qkv = rnd(B * S, 3 * H * hd, dtype=torch.bfloat16, seed=S + 7)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8)
        km[-1, S - S // 5:] = 0
    ref, lse_ref = _ref_attention(qkv, B, S, H, hd, hd ** -0.5, causal, km)
    out = torch.zeros(B * S, H * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    ops.attn_fwd(qkv.cuda(), B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=None if km is None else km.cuda())
    close(out, ref, 2e-2)
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    out2 = torch.zeros_like(out)
    ops.attn_fwd(qkv.cuda(), B, S, H, hd, hd ** -0.5, out2, None, causal=causal, key_mask=None if km is None else km.cuda())   # LSE optional
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,S,H,causal,masked,hd", [(2, 200, 3, True, True, 128), (1, 692, 2, True, False, 128), (2, 64, 1, False, True, 128),
                                                       (1, 33, 2, True, False, 128), (1, 300, 1, True, True, 128),
                                                       # head_dim 64: the PointBERT blocks under --unfreeze_pc_encoder (S = 513, 6 heads, no mask, not causal)
                                                       (2, 513, 6, False, False, 64), (1, 200, 2, False, True, 64), (2, 33, 1, True, False, 64), (1, 130, 3, True, True, 64)])
def test_fused_attention_backward(ops, B, S, H, causal, masked, hd):
    d = H * hd
    qkv = rnd(B * S, 3 * d, dtype=torch.bfloat16, seed=S + 1)
    dout = rnd(B * S, d, dtype=torch.bfloat16, seed=S + 2)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8)
        km[-1, S - S // 5:] = 0
    x = qkv.float().requires_grad_(True)
    ref, _ = _ref_attention(x, B, S, H, hd, hd ** -0.5, causal, km)
    ref.backward(dout.float())
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    Q = qkv.cuda()
    kmc = None if km is None else km.cuda()
    ops.attn_fwd(Q, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=kmc)
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device="cuda")
    delta = torch.zeros(B, H, S, dtype=torch.float32, device="cuda")
    ops.attn_bwd(Q, out, lse, dout.cuda(), dqkv, delta, B, S, H, hd, hd ** -0.5, causal=causal, key_mask=kmc)
    g = x.grad
    for i, name in enumerate("qkv"):
        got, want = dqkv[:, i * d:(i + 1) * d], g[:, i * d:(i + 1) * d]
        err = (got.float().cpu() - want).abs().max().item()
        assert err <= 3e-2 * (want.abs().max().item() + 1e-9), (name, err, want.abs().max().item())


@pytest.mark.parametrize("B,S,H,causal,mask,rope", [(2, 692, 3, True, "tail", True), (1, 692, 2, True, None, False), (1, 200, 2, True, "holes", True),
                                                    (2, 64, 1, False, "tail", False), (1, 33, 2, True, None, True), (1, 1, 1, True, None, False),
                                                    (2, 300, 2, False, "holes", False), (1, 1000, 1, True, "tail", True),
                                                    # S % 32 == 0 over several tiles: no clamped last tile anywhere (round 3: the dK/dV kernel's LSE / delta
                                                    # fetch of the last tile ran 128 B past the end of those arrays here; S = 256 was the first such S in the suite)
                                                    (8, 256, 16, True, "tail", True), (2, 512, 2, False, None, False), (1, 128, 1, True, None, True)])
def test_fused_attention_backward_second_form_is_bit_identical(ops, B, S, H, causal, mask, rope):
    """attn_bwd_dq2_kernel / attn_bwd_dkdv2_kernel against the first forms: the same arithmetic in the same order -> the same bits in dq, dk,
    dv and delta (interior, diagonal, ragged and padded tiles, dead waves, with and without the inverse-RoPE epilogue)."""
    from egoscaler_amd import _lib
    L = _lib.lib()
    hd = 128
    d = H * hd
    qkv = rnd(B * S, 3 * d, dtype=torch.bfloat16, seed=S + 11).cuda()
    dout = rnd(B * S, d, dtype=torch.bfloat16, seed=S + 12, scale=0.1).cuda()
    km = None
    if mask is not None:
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "tail":
            km[-1, S - max(1, S // 5):] = 0
        else:
            km[0, 2:4] = 0
            km[-1, S // 2] = 0
        km = km.cuda()
    out = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    # LSE and delta sit at the very END of allocations of their own (16 MB: a segment of its own in the caching allocator), so that a read past
    # their last element leaves the mapped segment instead of landing in a neighbouring tensor
    def at_end(n):
        buf = torch.zeros((16 << 20) // 4, dtype=torch.float32, device="cuda")
        return buf[buf.numel() - n:].view(B, H, S), buf
    lse, _keep0 = at_end(B * H * S)
    ops.attn_fwd(qkv, B, S, H, hd, hd ** -0.5, out, lse, causal=causal, key_mask=km)
    rp = None
    if rope:
        cos, sin = ops.rope_tables(max(S, 8), hd, 10000.0)
        rp = (cos.cuda(), sin.cuda())
    res = {}
    try:
        for form in (1, 2, 3):
            assert L.egomi_attn_set_bwd_form(form) == 0
            dqkv = torch.full((B * S, 3 * d), 3.0, dtype=torch.bfloat16, device="cuda")
            delta, _keep1 = at_end(B * H * S)
            delta.fill_(3.0)
            ops.attn_bwd(qkv, out, lse, dout, dqkv, delta, B, S, H, hd, hd ** -0.5, causal=causal, key_mask=km, rope=rp)
            torch.cuda.synchronize()
            res[form] = (dqkv, delta)
    finally:
        L.egomi_attn_set_bwd_form(3)
    for f in (2, 3):                                                        # form 3 (round 4): the dQ kernel on the forward's 32-key ring, end-aligned query blocks
        assert torch.equal(res[1][1], res[f][1]), f
        for i, nm in enumerate("qkv"):
            assert torch.equal(res[1][0][:, i * d:(i + 1) * d], res[f][0][:, i * d:(i + 1) * d]), (f, nm)
        assert bool(torch.isfinite(res[f][0].float()).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(ops, dtype):
    x = rnd(333, 517, dtype=dtype)
    out = torch.ones(517, dtype=torch.float32, device="cuda")
    ops.colsum_(x.cuda(), out)
    close(out, x.float().sum(0) + 1.0, 1e-4 if dtype == torch.float32 else 1e-3)
