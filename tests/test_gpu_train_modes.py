"""GPU: bf16 training modes through the whole step (loss_and_backward + EgoAdamW), including the
transposed-operand wgrad path and the W^T refresh after an optimizer step (unfrozen LLM)."""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu


def _model(dims, unfreeze, dtype):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=unfreeze, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=dtype)
    sd = synth.synth_state_dict(dims, 0)
    m.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
    return m


def _dims():
    d = dims_tiny()
    d.lm.hidden_size, d.lm.num_attention_heads, d.lm.intermediate_size = 256, 2, 512
    return d


@pytest.mark.parametrize("ops_dtype", [torch.bfloat16])
def test_transpose_vec_path(ops_dtype):
    from egoscaler_amd import ops
    for R, C, ldo in [(700, 4096, 704), (64, 128, 64), (130, 72, 192), (5, 8, 8)]:
        x = torch.randn(R, C).to(ops_dtype)
        t = ops.transpose(x.cuda(), ldo=ldo)
        assert torch.equal(t[:, :R].cpu(), x.t()) and (ldo == R or float(t[:, R:].abs().max()) == 0)
    wide = torch.randn(100, 256).to(ops_dtype).cuda()
    t = ops.transpose(wide[:, 64:192], ldo=128)                         # column-slice view (ldi > C)
    assert torch.equal(t[:, :100].cpu(), wide[:, 64:192].cpu().t())


def test_bf16_unfrozen_grads_track_fp32_and_training_reduces_loss():
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 17, 3, 9]
    ref = _model(dims, True, torch.float32)
    ref.train()
    l32 = ref.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
    g32 = {n: p.main_grad.clone() for n, p in ref.named_parameters() if getattr(p, "main_grad", None) is not None}
    m = _model(dims, True, torch.bfloat16)
    m.train()
    l16 = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
    assert abs(float(l16) - float(l32)) < 2e-2 * abs(float(l32))
    for n, p in m.named_parameters():
        if n in g32:
            g = g32[n]
            err = float((p.main_grad - g).abs().max())
            assert err <= 0.2 * float(g.abs().max()) + 1e-6, (n, err, float(g.abs().max()))
    # a few optimizer steps: the loss must go down, and dgrad must keep using fresh W^T copies
    opt = EgoAdamW(m, lr=2e-3, weight_decay=0.0)
    losses = [float(l16)]
    for _ in range(4):
        opt.step()
        losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)))
    assert losses[-1] < losses[0] - 0.05, losses
    eng = m.engine
    for nm, wt in eng.wT.items():
        assert torch.equal(wt, eng.w[nm].t().contiguous()), nm          # transposed copies follow the updated weights


def test_frozen_mode_only_updates_trainable_tensors():
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    m = _model(dims, False, torch.bfloat16)
    m.train()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = EgoAdamW(m, lr=1e-2, weight_decay=0.0)
    m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=[0, 17])
    opt.step()
    for n, p in m.named_parameters():
        changed = not torch.equal(p.detach(), before[n])
        frozen = n.startswith(("model.layers.", "model.point_backbone."))
        assert changed != frozen or (not frozen and n.endswith("embed_tokens.weight")), n


def test_unfrozen_stacked_products_over_parameter_views():
    """--unfreeze_language_model, bf16: q|k|v and gate|up are allocated side by side (model_arch.py), so the engine's stacked operands
    are VIEWS of the parameters and the step runs one forward product, one data-gradient product (k-major kernel, K = 3d / 2*ffn) and one
    weight-gradient product (into the stacked view of the fp32 gradient block) per group.  Checked against the same model run with the
    separate products (the side-by-side detection switched off), and after an optimizer step (a view needs no refresh)."""
    from egoscaler_amd.engine import Engine
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    dims.lm.hidden_size, dims.lm.num_attention_heads, dims.lm.intermediate_size = 2048, 16, 2816
    B = 8
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=60, num_steps=20, max_traj_token=160)
    assert toks.shape[1] == 256
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)])
    start = list(range(B))

    def run(stacked):
        m = _model(dims, True, torch.bfloat16)
        m.train()
        eng = m.engine
        calls = {"dgrad": 0, "wgrad": 0}
        if not stacked:
            eng._side_by_side = lambda ts: None
        dg, wg = eng._dgrad_w, eng._wgrad_into

        def dgrad_w(dY, W, out, residual=None):
            calls["dgrad"] += 1
            return dg(dY, W, out, residual)

        def wgrad_into(g, acc, dY, X):
            calls["wgrad"] += 1
            return wg(g, acc, dY, X)
        eng._dgrad_w, eng._wgrad_into = dgrad_w, wgrad_into
        loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
        grads = {n: p.main_grad.clone() for n, p in m.named_parameters() if getattr(p, "main_grad", None) is not None}
        return m, float(loss), grads, calls

    m, loss_s, g_s, c_s = run(True)
    eng, L, d = m.engine, dims.lm.num_hidden_layers, dims.lm.hidden_size
    for l in range(L):
        q = eng.w[f"model.layers.{l}.self_attn.q_proj.weight"]
        assert eng.wqkv[l].data_ptr() == q.data_ptr() and eng.wqkv[l].shape == (3 * d, d)
        assert eng.wgu_cat[l].data_ptr() == eng.w[f"model.layers.{l}.mlp.gate_proj.weight"].data_ptr()
        assert torch.equal(eng.wqkv[l][d:2 * d], eng.w[f"model.layers.{l}.self_attn.k_proj.weight"])
        assert torch.equal(eng.wgu_cat[l][dims.lm.intermediate_size:], eng.w[f"model.layers.{l}.mlp.up_proj.weight"])
    _, loss_p, g_p, c_p = run(False)
    # per layer: stacked = 4 data-gradient products (down, gate|up, o, q|k|v) against 7; weight gradients 2 (down, o) through the
    # per-name route + 2 stacked ones that bypass it, against 7
    assert c_s["dgrad"] == 4 * L and c_p["dgrad"] == 7 * L, (c_s, c_p)
    assert c_p["wgrad"] - c_s["wgrad"] == 5 * L, (c_s, c_p)
    assert abs(loss_s - loss_p) <= 2e-3 * abs(loss_p), (loss_s, loss_p)
    assert g_s.keys() == g_p.keys()
    for n in g_s:
        ref = g_p[n]
        err = float((g_s[n] - ref).abs().max())
        assert err <= 3e-2 * float(ref.abs().max()) + 1e-7, (n, err, float(ref.abs().max()))
    opt = EgoAdamW(m, lr=1e-3, weight_decay=0.0)
    before = eng.wqkv[0].clone()
    opt.step()
    assert not torch.equal(before, eng.wqkv[0])                              # the stack IS the parameters: it moved with them
    assert torch.equal(eng.wqkv[0][2 * d:], dict(m.named_parameters())["model.layers.0.self_attn.v_proj.weight"].data)
    l2 = float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start))
    assert l2 < loss_s, (loss_s, l2)
    sd = m.state_dict()
    assert sd["model.layers.1.self_attn.k_proj.weight"].shape == (d, d) and sd["model.layers.1.mlp.up_proj.weight"].is_contiguous()


@pytest.mark.parametrize("unfreeze", [False, True])
def test_overlapped_optimizer_step_equals_the_plain_one(unfreeze):
    """EgoAdamW.step(overlap=True): the updates run on a side stream in the order the next forward pass reads the tensors, the forward waits per
    group (Engine.wait_params).  Same losses and the same weights as the plain step over three steps; reading through state_dict() / generate()
    waits for the updates in flight."""
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 17, 3, 9]
    res = {}
    for overlap in (False, True):
        m = _model(dims, unfreeze, torch.bfloat16)
        m.train()
        opt = EgoAdamW(m, lr=2e-3, weight_decay=0.01)
        losses = []
        for _ in range(3):
            losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)))
            opt.step(overlap=overlap)
        if overlap:
            assert m.engine.param_events                      # updates of the last step are still registered ...
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}       # ... and state_dict() waits for them
        assert not m.engine.param_events
        st = opt.state_dict_cpu()
        res[overlap] = (losses, sd, st)
        if overlap:                                           # generation right after an overlapped step reads finished weights
            opt.step(overlap=True)
            out = m.generate(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), max_length=3, do_sample=False,
                             fps_start=start)
            assert out.sequences.shape == (4, Lp + 3) and not m.engine.param_events
    (la, sa, oa), (lb, sb, ob) = res[False], res[True]
    # round 4: no reduction of the step meets through fp32 atomics any more (loss, norm-weight and embedding gradients are ordered sums), so a
    # stream-ordering bug — a forward reading a half-updated group, a gradient overwritten under the update — cannot hide inside a tolerance:
    # losses, every weight and every optimizer moment are bit-identical between the two schedules
    assert la == lb, (la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for n in oa["state"]:
        for part in ("master", "m", "v"):
            assert torch.equal(oa["state"][n][part], ob["state"][n][part]), (n, part)
    assert oa["t"] == ob["t"] == 3


def test_armed_optimizer_updates_layers_under_the_backward_pass_and_equals_the_plain_step():
    """EgoAdamW.arm() (round 4): with every decoder layer trainable and one rank, layer l is updated on the side stream as soon as ITS
    gradients are final — under the backward pass of the layers below — and step(overlap=True) finishes the rest.  Same losses, weights and
    optimizer moments, bit for bit, as the plain step over three steps; the hook really ran for every layer; with lr / grad_scale that differ
    from arm()'s, step() refuses; frozen decoder layers arm nothing."""
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 17, 3, 9]
    res = {}
    for early in (False, True):
        m = _model(dims, True, torch.bfloat16)
        m.train()
        opt = EgoAdamW(m, lr=2e-3, weight_decay=0.01)
        losses, seen = [], []
        for i in range(3):
            if early:
                assert opt.arm(grad_scale=0.5, lr=1e-3 * (i + 1))
                hook = m.engine.layer_final_hook
                m.engine.layer_final_hook = lambda l, hook=hook: (seen.append(l), hook(l))[1]
            losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)))
            opt.step(grad_scale=0.5, lr=1e-3 * (i + 1), overlap=early)
            assert m.engine.layer_final_hook is None
        if early:
            L = dims.lm.num_hidden_layers
            assert seen == list(reversed(range(L))) * 3, seen
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        res[early] = (losses, sd, opt.state_dict_cpu())
        if early:
            opt.arm(grad_scale=0.5, lr=1e-3)
            m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
            with pytest.raises(RuntimeError):
                opt.step(grad_scale=1.0, lr=1e-3, overlap=True)
    (la, sa, oa), (lb, sb, ob) = res[False], res[True]
    assert la == lb, (la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for n in oa["state"]:
        for part in ("master", "m", "v"):
            assert torch.equal(oa["state"][n][part], ob["state"][n][part]), (n, part)
    mf = _model(dims, False, torch.bfloat16)
    assert EgoAdamW(mf, lr=1e-3).arm() is False and mf.engine.layer_final_hook is None


@pytest.mark.parametrize("unfreeze", [False, True])
def test_zero_grad_after_an_overlapped_step_waits_for_the_update(unfreeze):
    """The reference loop's order (train.py:159-184: zero_grad at the top of the next iteration, after step) with step(overlap=True): the
    side-stream AdamW kernels still read main_grad when zero_grad() clears it on the compute stream — Engine.zero_grad makes the compute
    stream wait for the updates in flight first.  Weights, moments and losses equal the plain step's bit for bit."""
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 17, 3, 9]
    res = {}
    for overlap in (False, True):
        m = _model(dims, unfreeze, torch.bfloat16)
        m.train()
        opt = EgoAdamW(m, lr=2e-3, weight_decay=0.01)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)))
            opt.step(overlap=overlap)
        opt.zero_grad()                                       # right behind the last (possibly still running) update
        assert not m.engine.param_events
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        res[overlap] = (losses, sd, opt.state_dict_cpu())
    (la, sa, oa), (lb, sb, ob) = res[False], res[True]
    assert la == lb, (la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for n in oa["state"]:
        for part in ("master", "m", "v"):
            assert torch.equal(oa["state"][n][part], ob["state"][n][part]), (n, part)


def test_local_bf16_wire_gradients_on_one_rank_train_like_the_fp32_buffers():
    """dp.GradSync(resident=True, local=True) on ONE rank (round 4): no exchange, but every decoder layer's weight gradients are produced in bf16 inside
    the layer's wire buffer and read there by EgoAdamW — what a DP job (tests/test_gpu_dp.py) and the reference's DeepSpeed bf16 engine train on — also
    with the armed per-layer updates.  Three steps: the losses follow the fp32-gradient run closely (bf16 rounding of the gradients only), the wire
    buffers were really used, and the weights moved."""
    from egoscaler_amd.dp import GradSync
    from egoscaler_amd.optim import EgoAdamW
    dims = _dims()
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 17, 3, 9]
    res = {}
    for local in (False, True):
        m = _model(dims, True, torch.bfloat16)
        m.train()
        opt = EgoAdamW(m, lr=1e-3, weight_decay=0.01)
        sync = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1 << 10, resident=True, local=True) if local else None
        m.engine.grad_sync = sync
        losses, used = [], 0
        for i in range(3):
            assert opt.arm(grad_scale=1.0) is True
            losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)))
            used += sum(1 for v in m.engine.reduced_grad.values() if v.dtype == torch.bfloat16)
            if sync is not None:
                sync.finish()
            opt.step(grad_scale=1.0, overlap=True)
        assert (used > 0) == local, used
        res[local] = (losses, {k: v.detach().float().clone() for k, v in m.state_dict().items() if "layers.0.mlp.down_proj" in k or "layers.1.self_attn.q_proj" in k})
    (la, wa), (lb, wb) = res[False], res[True]
    assert la[0] == lb[0]                                      # the first forward is the same computation
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-2 * abs(x), (la, lb)
    for k in wa:
        assert float((wa[k] - wb[k]).abs().max()) <= 5e-2 * float(wa[k].abs().max()) and not torch.equal(wa[k], wb[k]), k
