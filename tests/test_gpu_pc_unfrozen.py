"""GPU: --unfreeze_pc_encoder (point backbone trainable, train() mode: BatchNorm on batch statistics,
running-stat update) against golden vectors recorded from the reference in that mode
(tests/golden/tiny_pc_unfrozen.npz; DropPath rate 0 — stochastic depth cannot be pinned)."""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu
REL = 1e-3


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _model(dims, dtype=torch.float32):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    args = types.SimpleNamespace(unfreeze_pc_encoder=True, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=dtype)
    sd = synth.synth_state_dict(dims, 0)
    m.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
    return m


def test_unfrozen_point_backbone_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_pc_unfrozen.npz"), allow_pickle=False)
    dims = dims_tiny()
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    m = _model(dims)
    m.train()
    assert m.model.point_backbone.training and m.engine.pb_train_mode
    loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=g["fps_start"])
    assert abs(float(loss) - float(g["loss"])) < REL * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    got = sorted(n for n, p in params.items() if getattr(p, "main_grad", None) is not None)
    assert got == g["grad_names_all"].tolist()
    for k in g.files:
        if k.startswith("grad:"):
            ref, mine = g[k], params[k[5:]].main_grad
            if np.abs(ref).max() < 1e-6:          # conv bias in front of a train-mode BatchNorm: exact gradient is zero
                assert float(mine.abs().max()) < 1e-5, k
            else:
                assert rel(mine, ref) < REL, (k, rel(mine, ref))
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("after:"):
            ref = g[k]
            if ref.dtype.kind in "iu":
                assert int(sd[k[6:]]) == int(ref)
            else:
                assert rel(sd[k[6:]], ref) < REL, k
    # eval() afterwards re-folds BatchNorm with the UPDATED running statistics
    from oracle import pointllm as OPL
    m.eval()
    with torch.no_grad():
        lg = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts.cuda(), fps_start=g["fps_start"]).logits
    sdo = {k: v.detach().cpu().clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = OPL.forward(sdo, dims, toks, masks, pts, g["fps_start"])
    assert rel(lg, ref.numpy()) < REL


def test_unfrozen_point_backbone_trains_with_droppath():
    from egoscaler_amd.optim import EgoAdamW
    dims = dims_tiny()
    dims.pb.drop_path_rate = 0.2
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    m = _model(dims)
    m.train()
    opt = EgoAdamW(m, lr=2e-3, weight_decay=0.0)
    before = m.state_dict()["model.point_backbone.blocks.blocks.1.mlp.fc1.weight"].clone()
    torch.manual_seed(0)
    losses = []
    for _ in range(6):
        losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=[0, 1, 2, 3])))
        opt.step()
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert not torch.equal(before, m.state_dict()["model.point_backbone.blocks.blocks.1.mlp.fc1.weight"])
    sc = m.engine.pb_trainer.drop_scales(4)
    assert sc.shape == (dims.pb.depth, 2, 4) and float(sc[0].min()) == 1.0 and set(np.unique(sc[-1].cpu().numpy()).round(4)) <= {0.0, 1.25}


def test_bf16_point_backbone_training_on_fused_head_dim_64_attention():
    """--unfreeze_pc_encoder in bf16 with PointBERT heads of 64 (the real PointBERT-v1.2 has 6 x 64): the blocks run the fused attention forward
    (LSE kept) and the fused head_dim-64 backward kernels instead of materialised [B*H, S, S] probabilities.  Against the same model on the
    unfused path (bf16), and against the fp32 engine whose gradients the golden test above pins to the reference."""
    dims = dims_tiny()
    dims.pb.trans_dim, dims.pb.num_heads = 128, 2                    # head_dim 64; 33 tokens per cloud (ragged 32-row tiles)
    dims.pb.projection_hidden_dim = [64, 96]
    toks, masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(4)])
    start = [0, 5, 9, 2]
    res = {}
    for key, dtype, fused in (("fused", torch.bfloat16, True), ("unfused", torch.bfloat16, False), ("fp32", torch.float32, False)):
        m = _model(dims, dtype)
        m.train()
        m.engine.use_fused_attention = fused
        calls = []
        from egoscaler_amd import ops
        orig = ops.attn_bwd

        def spy(*a, **k):
            calls.append(a[9])                                           # head_dim
            return orig(*a, **k)
        ops.attn_bwd = spy
        try:
            loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
        finally:
            ops.attn_bwd = orig
        res[key] = (float(loss), {n: p.main_grad.float().clone() for n, p in m.named_parameters()
                                  if n.startswith("model.point_backbone.blocks.") and getattr(p, "main_grad", None) is not None}, calls)
    assert res["fused"][2].count(64) == dims.pb.depth and 64 not in res["unfused"][2]
    assert abs(res["fused"][0] - res["fp32"][0]) < 2e-2 * abs(res["fp32"][0])
    worst = 0.0
    for n, g32 in res["fp32"][1].items():
        scale = float(g32.abs().max()) + 1e-12
        e_f = float((res["fused"][1][n] - g32).abs().max()) / scale
        e_u = float((res["unfused"][1][n] - g32).abs().max()) / scale
        worst = max(worst, e_f)
        assert e_f <= max(2.0 * e_u, 5e-2), (n, e_f, e_u)                # as close to fp32 as the unfused bf16 path is
    assert worst > 0


def test_unfrozen_point_backbone_with_droppath_matches_reference(golden_dir):
    """Stochastic depth ON (point_encoder.py:65,74-75,133-134): the reference's run took its per-sample DropPath draws from
    the fixture generator (oracle/_shims/timm: timm 0.4.12's x / keep * floor(keep + U) with U handed in; depth 2 at rate 0.5 -> block 1 drops
    at p = 0.5, block 0 is an Identity), the product receives the same draws as branch scales [depth, 2, B].  Loss, logits-derived loss and
    the point-backbone gradients <= 1e-3; the rate-0 loss of the same batch differs (the draws matter)."""
    g = np.load(os.path.join(golden_dir, "tiny_pc_unfrozen_droppath.npz"), allow_pickle=False)
    dims = dims_tiny()
    dims.pb.drop_path_rate = float(g["drop_path_rate"])
    B = 4
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)])
    m = _model(dims)
    m.train()
    scales = torch.from_numpy(g["drop_scales"]).to(torch.float32)
    assert scales.shape == (dims.pb.depth, 2, B) and float(scales[0].min()) == 1.0 and sorted(set(scales[1].flatten().tolist())) == [0.0, 2.0]
    m.engine.pb_drop_override = scales.cuda().contiguous()
    loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=g["fps_start"])
    assert m.engine.pb_drop_override is None                              # consumed by that pass
    assert abs(float(loss) - float(g["loss"])) < REL * abs(float(g["loss"]))
    assert abs(float(g["loss_rate0"]) - float(g["loss"])) > 1e-4 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    got = sorted(n for n, p in params.items() if getattr(p, "main_grad", None) is not None)
    assert got == g["grad_names_all"].tolist()
    for k in g.files:
        if k.startswith("grad:"):
            assert rel(params[k[5:]].main_grad, g[k]) < REL, (k, rel(params[k[5:]].main_grad, g[k]))


def test_point_backbone_with_96_wide_heads_matches_oracle():
    """The second PointBERT YAML the reference ships (PointTransformer_base_8192point.yaml: trans_dim 1152 = 12 heads x 96, no projection hidden
    layers; accepted by PointLLMConfig.POINTBERT_BY_NAME) at tiny size: head_dim 96 has no fused attention kernel (engine.py fuses 64 and 128),
    so the blocks take the generic batched-GEMM + softmax path — pinned here against the oracle's run of the same geometry."""
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    from oracle import pointllm as OPL
    dims = dims_tiny()
    dims.pb.trans_dim, dims.pb.num_heads = 192, 2                          # head_dim 96
    dims.pb.projection_hidden_dim = []                                     # projection_hidden_layer: 0 -> one Linear (pointllm.py:78-81)
    assert dims.pb.head_dim == 96
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32).eval()
    sd = synth.synth_state_dict(dims, 0)
    assert sd["model.point_backbone.blocks.blocks.0.attn.qkv.weight"].shape == (576, 192) and "model.point_proj.2.weight" not in sd
    m.load_state_dict(sd)
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    with torch.no_grad():
        lg = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts.cuda(), fps_start=[0, 17]).logits
        ref = OPL.forward({k: v.clone() for k, v in sd.items()}, dims, toks, masks, pts, np.array([0, 17]))
    assert rel(lg, ref.numpy()) < REL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unfrozen_point_backbone_step_is_replayable_bit_for_bit(dtype):
    """Two fresh models, the same batch: loss, every gradient (the point backbone's included) and the BatchNorm running statistics are the
    SAME BITS.  The column reductions of csrc/pointbert_train.hip (LayerNorm dw / db, train-mode BatchNorm batch statistics and dgamma /
    dbeta, the small-K weight gradients) are ordered two-stage sums since round 4; before, fp32 atomics made even the forward differ in
    its last bits (the batch statistics).  DropPath off (its masks are fresh draws)."""
    dims = dims_tiny()
    dims.pb.drop_path_rate = 0.0
    B = 4
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)])
    start = torch.zeros(B, dtype=torch.int32)
    runs = []
    for _ in range(2):
        m = _model(dims, dtype)
        m.train()
        loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start.cuda())
        torch.cuda.synchronize()
        grads = {n: p.main_grad.clone() for n, p in m.named_parameters() if getattr(p, "main_grad", None) is not None}
        stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k}
        runs.append((float(loss), grads, stats))
        del m
    assert runs[0][0] == runs[1][0]
    assert any(n.startswith("model.point_backbone.") for n in runs[0][1])
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k
