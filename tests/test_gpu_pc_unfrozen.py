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
