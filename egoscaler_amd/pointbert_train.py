"""Trainable point backbone (--unfreeze_pc_encoder): forward in train mode + full backward.

Reference modules: pointbert/dvae.py:150-221 (Group, Encoder with BatchNorm1d in TRAIN mode: batch
statistics over all B*G*M rows, running-stat update), pointbert/point_encoder.py:11-98,169-189
(ViT blocks with DropPath, pos re-added before every block, final LayerNorm).  The frozen / eval path
stays in Engine.point_backbone (BatchNorm folded into the convs); this module is used only when the
backbone's parameters are trainable and the model is in train() mode (model_arch.py:33-36,110-124).
Gradients go to the engine's fp32 main_grad buffers.  No gradient flows to the input points.
"""
import torch

from . import ops
from ._lib import c_i, c_f, c_i64, call
from .ops import P, S, dt

PRE = "model.point_backbone."


_PART = {}


def _partials(n, device):
    """fp32 scratch for the ordered two-stage column reductions (include/egomi.h): grow-only, one per device; every user runs on the
    current stream, so consecutive calls may share it."""
    t = _PART.get(device)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        _PART[device] = t
    return t


def layernorm_bwd(dy, x, w, eps, dx_add=None, dw=None, db=None, out=None):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    pt = _partials(min((rows + 3) // 4, 512) * 2 * cols, x.device)
    call("egomi_layernorm_bwd", P(dy), P(x), P(w), P(out), P(dx_add), P(dw), P(db), c_i(rows), c_i(cols), c_f(eps), P(pt), c_i64(pt.numel()),
         c_i(dt(x.dtype)), S())
    return out


def bn_train_fwd(x, gamma, beta, eps, relu, stats, rmean, rvar, momentum=0.1, out=None):
    R, C = x.shape
    out = torch.empty_like(x) if out is None else out
    pt = _partials((R + 255) // 256 * 2 * C, x.device)
    call("egomi_bn_train_fwd", P(x), c_i64(R), c_i(C), P(gamma), P(beta), c_f(eps), c_i(int(relu)), P(out), P(stats), P(rmean), P(rvar),
         c_f(momentum), P(pt), c_i64(pt.numel()), c_i(dt(x.dtype)), S())
    return out


def bn_train_bwd(dy, x, y, stats, gamma, relu, dgamma, dbeta, out=None):
    R, C = x.shape
    out = torch.empty_like(x) if out is None else out
    pt = _partials((R + 255) // 256 * 2 * C, x.device)
    call("egomi_bn_train_bwd", P(dy), P(x), P(y), c_i64(R), c_i(C), P(stats), P(gamma), c_i(int(relu)), P(dgamma), P(dbeta), P(out),
         P(pt), c_i64(pt.numel()), c_i(dt(x.dtype)), S())
    return out


def group_argmax(x, BG, M, C):
    out = torch.empty(BG, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(BG, C, dtype=torch.int32, device=x.device)
    call("egomi_group_argmax", P(x), c_i(BG), c_i(M), c_i(C), P(out), P(idx), c_i(dt(x.dtype)), S())
    return out, idx


def group_max_bwd(dout, idx, BG, M, C, dx, accumulate):
    call("egomi_group_max_bwd", P(dout), P(idx), c_i(BG), c_i(M), c_i(C), P(dx), c_i64(dx.stride(0)), c_i(int(accumulate)), c_i(dt(dout.dtype)), S())
    return dx


def smallk_wgrad(dy, x, dW):
    R, N = dy.shape
    K = x.shape[-1]
    pt = _partials((R + 511) // 512 * N * K, dy.device)
    call("egomi_smallk_wgrad", P(dy), P(x), c_i(dt(x.dtype)), c_i64(R), c_i(N), c_i(K), P(dW), P(pt), c_i64(pt.numel()), c_i(dt(dy.dtype)), S())


def rowscale_add(resid, branch, scale, rows_per_sample, out=None):
    rows, cols = branch.shape
    out = torch.empty_like(branch) if out is None else out
    call("egomi_rowscale_add", P(resid), P(branch), P(scale), c_i64(rows), c_i(cols), c_i(rows_per_sample), P(out), c_i(dt(branch.dtype)), S())
    return out


class PointBackboneTrainer:
    def __init__(self, engine):
        self.eng = engine

    # ------------------------------------------------------------------------------------ helpers
    def _g(self, name):
        return self.eng.grad_buffer(PRE + name)

    def _wg(self, name, dY, X):
        """g[N, K] += dY^T [N, R] . X [R, K].  The outputs are small (2 - 36 tiles of 128 x 128) and the reduction long (R = B*G*M = 131072 rows in the
        mini-PointNet, B*513 in the blocks): one block per tile walked the whole reduction alone (8.7 ms for the 256 x 128 weight: `bench.py --mode pc`
        spent 41 ms per step here).  The reduction is cut into S equal row ranges computed as ONE batched launch into fp32 partials, summed in slice
        order (egomi_rank_sum) and added to the gradient: deterministic, ~S x the blocks."""
        g = self._g(name)
        g2 = g.view(g.shape[0], -1)
        R, N = dY.shape
        K = X.shape[1]
        if self.eng.dtype == torch.bfloat16 and 128 <= R <= 16384 and N * K >= 128 * 128:
            # the transformer blocks' linears (R = B*513): both operands transposed into zero-padded [*, R64] buffers and the K-contiguous tuned
            # kernel with its own split-K (engine._wgrad_into): 40 vs 80 us per weight on the generic k-major kernel
            self.eng._wgrad_into(g2, True, dY, X)
            return
        tiles = -(-N // 128) * -(-K // 128)
        S = 1
        for cand in (128, 64, 32, 16, 8, 4, 2):
            if tiles * cand <= 1024 and R % cand == 0 and R // cand >= 256:
                S = cand
                break
        if S == 1 or (N * K) % 8:
            ops.mm(dY, X, out=g2, a_layout=1, b_layout=1, accumulate=True)
            return
        chunk = R // S
        part = self.eng.ws.get(f"pbwg_part_{S}_{N}_{K}", (S, N * K), torch.float32)
        ops.gemm_raw(dY, X, part, N, K, chunk, ops._ld(dY), ops._ld(X), K, 1, 1, batch=S,
                     strides=(chunk * ops._ld(dY), 0, chunk * ops._ld(X), 0, N * K, 0))
        tot = self.eng.ws.get(f"pbwg_sum_{N}_{K}", (N * K,), torch.float32)
        ops.rank_sum(part, tot)
        ops.add(g2.reshape(-1), tot, out=g2.reshape(-1))

    def _bg(self, name, dY):
        ops.colsum_(dY, self._g(name))

    def _dgrad(self, dY, W2d):
        """dX = dY . W.  bf16: the (small, trainable) weight is transposed on the fly so that the product runs K-contiguous on the tuned kernels
        (a <= 1.2-MB pass against 62 us per product on the generic kernel)."""
        N, K = W2d.shape
        if self.eng.dtype == torch.bfloat16 and N % 64 == 0 and K % 8 == 0 and W2d.is_contiguous():
            wt = ops.transpose(W2d, out=self.eng.ws.get(f"pb_wT_{N}_{K}", (K, N), self.eng.dtype))
            return ops.mm(dY, wt)
        return ops.mm(dY, W2d, b_layout=1)

    def drop_scales(self, B):
        """timm DropPath (point_encoder.py:65): per sample and branch, scale = floor(keep + U[0,1)) / keep,
        rates linspace(0, drop_path_rate, depth) (point_encoder.py:133)."""
        pb = self.eng.dims.pb
        rate = float(getattr(pb, "drop_path_rate", 0.0))
        if rate <= 0.0:
            return None
        dpr = torch.linspace(0, rate, pb.depth)
        keep = (1.0 - dpr)[:, None, None]
        mask = torch.floor(keep + torch.rand(pb.depth, 2, B))
        return (mask / keep).to(torch.float32).to(self.eng.device).contiguous()

    # ------------------------------------------------------------------------------------ forward
    def forward(self, pts, fps_start, drop=None):
        eng = self.eng
        w, pb, T = eng.w, eng.dims.pb, eng.dtype
        dev = eng.device
        B, N, C = pts.shape
        G, K, D, Pn = pb.num_group, pb.group_size, pb.trans_dim, pb.point_token_len
        H, hd = pb.num_heads, pb.head_dim
        BG, R, M = B * G, B * G * K, B * Pn
        c = {"B": B, "drop": drop}
        _, center = ops.fps(pts, fps_start, G)
        _, nb = ops.knn_group(pts, center, K, out_dtype=T)
        X0 = nb.view(R, C)
        e = PRE + "encoder."
        W1, W2 = w[e + "first_conv.0.weight"].view(pb.pn_c1, C), w[e + "first_conv.3.weight"].view(pb.pn_c2, pb.pn_c1)
        W3, W4 = w[e + "second_conv.0.weight"].view(pb.pn_c3, 2 * pb.pn_c2), w[e + "second_conv.3.weight"].view(pb.encoder_dims, pb.pn_c3)
        h1p = ops.linear_smallk(X0, W1, w[e + "first_conv.0.bias"])
        st1 = torch.empty(4 * pb.pn_c1, dtype=torch.float32, device=dev)
        y1 = bn_train_fwd(h1p, w[e + "first_conv.1.weight"], w[e + "first_conv.1.bias"], pb.bn_eps, True, st1,
                          w[e + "first_conv.1.running_mean"], w[e + "first_conv.1.running_var"])
        w[e + "first_conv.1.num_batches_tracked"] += 1
        h2 = ops.mm(y1, W2, bias=w[e + "first_conv.3.bias"])
        _, am2 = group_argmax(h2, BG, K, pb.pn_c2)
        cat = ops.group_max(h2, BG, K, pb.pn_c2, concat=True)
        h3p = ops.mm(cat, W3, bias=w[e + "second_conv.0.bias"])
        st3 = torch.empty(4 * pb.pn_c3, dtype=torch.float32, device=dev)
        y3 = bn_train_fwd(h3p, w[e + "second_conv.1.weight"], w[e + "second_conv.1.bias"], pb.bn_eps, True, st3,
                          w[e + "second_conv.1.running_mean"], w[e + "second_conv.1.running_var"])
        w[e + "second_conv.1.num_batches_tracked"] += 1
        h4 = ops.mm(y3, W4, bias=w[e + "second_conv.3.bias"])
        tok, am4 = group_argmax(h4, BG, K, pb.encoder_dims)
        c.update(X0=X0, center=center, h1p=h1p, st1=st1, y1=y1, am2=am2, cat=cat, h3p=h3p, st3=st3, y3=y3, am4=am4, tok=tok)
        x = torch.empty(B, Pn, D, dtype=T, device=dev)
        pos = torch.empty(B, Pn, D, dtype=T, device=dev)
        x[:, 0] = w[PRE + "cls_token"].view(1, D)
        pos[:, 0] = w[PRE + "cls_pos"].view(1, D)
        ops.gemm_raw(tok, w[PRE + "reduce_dim.weight"], x[:, 1:], G, D, pb.encoder_dims, pb.encoder_dims, pb.encoder_dims, D,
                     bias=w[PRE + "reduce_dim.bias"], batch=B, strides=(G * pb.encoder_dims, 0, 0, 0, Pn * D, 0))
        ph_pre = ops.linear_smallk(center.view(BG, 3), w[PRE + "pos_embed.0.weight"], w[PRE + "pos_embed.0.bias"])
        ph = ops.gelu(ph_pre)
        ops.gemm_raw(ph, w[PRE + "pos_embed.2.weight"], pos[:, 1:], G, D, pb.pos_hidden, pb.pos_hidden, pb.pos_hidden, D,
                     bias=w[PRE + "pos_embed.2.bias"], batch=B, strides=(G * pb.pos_hidden, 0, 0, 0, Pn * D, 0))
        c.update(ph_pre=ph_pre, ph=ph, blocks=[])
        xf, posf = x.view(M, D), pos.view(M, D)
        scale = hd ** -0.5
        fused = eng.use_fused_attention and T == torch.bfloat16 and hd == 64
        for i in range(pb.depth):
            p = f"{PRE}blocks.blocks.{i}."
            xs = torch.empty(M, D, dtype=T, device=dev)
            h = ops.layernorm(xf, w[p + "norm1.weight"], w[p + "norm1.bias"], pb.ln_eps, add=posf, sum_out=xs)
            qkv = ops.mm(h, w[p + "attn.qkv.weight"])
            ao = torch.empty(M, D, dtype=T, device=dev)
            Pm = lse = None
            if fused:
                # fused flash-style kernels at head_dim 64 (csrc/attention.hip): the [B*H, S, S] probabilities never reach HBM; LSE and the
                # output are kept for the backward kernels
                lse = torch.empty(B, H, Pn, dtype=torch.float32, device=dev)
                ops.attn_fwd(qkv, B, Pn, H, hd, scale, ao, lse, causal=False)
            else:
                Pm = eng._attention(qkv, B, Pn, H, hd, ao, False, None, scale, True)
            if drop is None:
                x1 = ops.mm(ao, w[p + "attn.proj.weight"], bias=w[p + "attn.proj.bias"], residual=xs)
            else:
                br = ops.mm(ao, w[p + "attn.proj.weight"], bias=w[p + "attn.proj.bias"])
                x1 = rowscale_add(xs, br, drop[i, 0], Pn)
            h2n = ops.layernorm(x1, w[p + "norm2.weight"], w[p + "norm2.bias"], pb.ln_eps)
            m_pre = ops.mm(h2n, w[p + "mlp.fc1.weight"], bias=w[p + "mlp.fc1.bias"])
            m = ops.gelu(m_pre)
            if drop is None:
                xn = ops.mm(m, w[p + "mlp.fc2.weight"], bias=w[p + "mlp.fc2.bias"], residual=x1)
            else:
                br = ops.mm(m, w[p + "mlp.fc2.weight"], bias=w[p + "mlp.fc2.bias"])
                xn = rowscale_add(x1, br, drop[i, 1], Pn)
            c["blocks"].append(dict(xs=xs, h=h, qkv=qkv, P=Pm, lse=lse, ao=ao, x1=x1, h2n=h2n, m_pre=m_pre, m=m))
            xf = xn
        out = ops.layernorm(xf, w[PRE + "norm.weight"], w[PRE + "norm.bias"], pb.ln_eps)
        c["x_last"] = xf
        return out.view(B, Pn, D), c

    # ------------------------------------------------------------------------------------ backward
    def backward(self, d_feats, c):
        eng = self.eng
        w, pb, T = eng.w, eng.dims.pb, eng.dtype
        dev = eng.device
        B = c["B"]
        G, K, D, Pn = pb.num_group, pb.group_size, pb.trans_dim, pb.point_token_len
        H, hd = pb.num_heads, pb.head_dim
        BG, R, M = B * G, B * G * K, B * Pn
        drop = c["drop"]
        scale = hd ** -0.5
        dx = layernorm_bwd(d_feats.reshape(M, D), c["x_last"], w[PRE + "norm.weight"], pb.ln_eps, dw=self._g("norm.weight"), db=self._g("norm.bias"))
        dpos = torch.zeros(M, D, dtype=T, device=dev)
        for i in reversed(range(pb.depth)):
            p = f"blocks.blocks.{i}."
            bc = c["blocks"][i]
            db2 = dx if drop is None else rowscale_add(None, dx, drop[i, 1], Pn)
            self._wg(p + "mlp.fc2.weight", db2, bc["m"])
            self._bg(p + "mlp.fc2.bias", db2)
            d_m = self._dgrad(db2, w[PRE + p + "mlp.fc2.weight"])
            d_mp = ops.gelu_bwd(d_m, bc["m_pre"])
            self._wg(p + "mlp.fc1.weight", d_mp, bc["h2n"])
            self._bg(p + "mlp.fc1.bias", d_mp)
            d_h2 = self._dgrad(d_mp, w[PRE + p + "mlp.fc1.weight"])
            d_x1 = layernorm_bwd(d_h2, bc["x1"], w[PRE + p + "norm2.weight"], pb.ln_eps, dx_add=dx,
                                 dw=self._g(p + "norm2.weight"), db=self._g(p + "norm2.bias"))
            db1 = d_x1 if drop is None else rowscale_add(None, d_x1, drop[i, 0], Pn)
            self._wg(p + "attn.proj.weight", db1, bc["ao"])
            self._bg(p + "attn.proj.bias", db1)
            d_ao = self._dgrad(db1, w[PRE + p + "attn.proj.weight"])
            dqkv = torch.empty(M, 3 * D, dtype=T, device=dev)
            if bc["lse"] is not None:
                ops.attn_bwd(bc["qkv"], bc["ao"], bc["lse"], d_ao, dqkv, eng.ws.get("pb_att_delta", (B, H, Pn), torch.float32), B, Pn, H, hd, scale,
                             causal=False)
            else:
                eng._attention_bwd(bc["qkv"], bc["P"], d_ao, dqkv, B, Pn, H, hd, scale)
            self._wg(p + "attn.qkv.weight", dqkv, bc["h"])
            d_h = self._dgrad(dqkv, w[PRE + p + "attn.qkv.weight"])
            dx = layernorm_bwd(d_h, bc["xs"], w[PRE + p + "norm1.weight"], pb.ln_eps, dx_add=d_x1,
                               dw=self._g(p + "norm1.weight"), db=self._g(p + "norm1.bias"))
            ops.add(dpos, dx, out=dpos)                       # block input was x + pos (point_encoder.py:95-98)
        # ---- tokens / positional embedding
        dx3, dp3 = dx.view(B, Pn, D), dpos.view(B, Pn, D)
        ops.colsum_(dx3.view(B, Pn * D)[:, :D], self._g("cls_token").view(-1))
        ops.colsum_(dp3.view(B, Pn * D)[:, :D], self._g("cls_pos").view(-1))
        d_tokD = dx3[:, 1:].contiguous().view(BG, D)
        d_posD = dp3[:, 1:].contiguous().view(BG, D)
        self._wg("reduce_dim.weight", d_tokD, c["tok"])
        self._bg("reduce_dim.bias", d_tokD)
        d_tok = self._dgrad(d_tokD, w[PRE + "reduce_dim.weight"])
        self._wg("pos_embed.2.weight", d_posD, c["ph"])
        self._bg("pos_embed.2.bias", d_posD)
        d_ph = self._dgrad(d_posD, w[PRE + "pos_embed.2.weight"])
        d_php = ops.gelu_bwd(d_ph, c["ph_pre"])
        smallk_wgrad(d_php, c["center"].view(BG, 3), self._g("pos_embed.0.weight"))
        self._bg("pos_embed.0.bias", d_php)
        # ---- mini-PointNet (dvae.py:207-221 backward)
        e = "encoder."
        C = c["X0"].shape[1]
        W2 = w[PRE + e + "first_conv.3.weight"].view(pb.pn_c2, pb.pn_c1)
        W3 = w[PRE + e + "second_conv.0.weight"].view(pb.pn_c3, 2 * pb.pn_c2)
        W4 = w[PRE + e + "second_conv.3.weight"].view(pb.encoder_dims, pb.pn_c3)
        d_h4 = torch.empty(R, pb.encoder_dims, dtype=T, device=dev)
        group_max_bwd(d_tok, c["am4"], BG, K, pb.encoder_dims, d_h4, False)
        self._wg(e + "second_conv.3.weight", d_h4, c["y3"])
        self._bg(e + "second_conv.3.bias", d_h4)
        d_y3 = self._dgrad(d_h4, W4)
        tg, tb = torch.empty(pb.pn_c3, dtype=torch.float32, device=dev), torch.empty(pb.pn_c3, dtype=torch.float32, device=dev)
        d_h3p = bn_train_bwd(d_y3, c["h3p"], c["y3"], c["st3"], w[PRE + e + "second_conv.1.weight"], True, tg, tb)
        ops.add(self._g(e + "second_conv.1.weight"), tg, out=self._g(e + "second_conv.1.weight"))
        ops.add(self._g(e + "second_conv.1.bias"), tb, out=self._g(e + "second_conv.1.bias"))
        self._wg(e + "second_conv.0.weight", d_h3p, c["cat"])
        self._bg(e + "second_conv.0.bias", d_h3p)
        d_cat = self._dgrad(d_h3p, W3)                                      # [R, 512] = [global | local]
        d_h2 = d_cat[:, pb.pn_c2:].contiguous()
        gsum = torch.empty(BG, pb.pn_c2, dtype=T, device=dev)              # expanded global feature: sum over the group's M rows
        call("egomi_group_sum", P(d_cat), c_i(BG), c_i(K), c_i(pb.pn_c2), c_i64(2 * pb.pn_c2), P(gsum), c_i(dt(T)), S())
        group_max_bwd(gsum, c["am2"], BG, K, pb.pn_c2, d_h2, True)
        self._wg(e + "first_conv.3.weight", d_h2, c["y1"])
        self._bg(e + "first_conv.3.bias", d_h2)
        d_y1 = self._dgrad(d_h2, W2)
        tg1, tb1 = torch.empty(pb.pn_c1, dtype=torch.float32, device=dev), torch.empty(pb.pn_c1, dtype=torch.float32, device=dev)
        d_h1p = bn_train_bwd(d_y1, c["h1p"], c["y1"], c["st1"], w[PRE + e + "first_conv.1.weight"], True, tg1, tb1)
        ops.add(self._g(e + "first_conv.1.weight"), tg1, out=self._g(e + "first_conv.1.weight"))
        ops.add(self._g(e + "first_conv.1.bias"), tb1, out=self._g(e + "first_conv.1.bias"))
        smallk_wgrad(d_h1p, c["X0"], self._g(e + "first_conv.0.weight"))
        self._bg(e + "first_conv.0.bias", d_h1p)
