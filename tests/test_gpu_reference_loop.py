"""GPU: the reference's own training-step idiom (train.py:159-184) works unchanged on the drop-in model
in fp32: logits slice -> F.cross_entropy -> loss.backward() -> a stock torch optimizer steps p.grad."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu


def test_reference_train_step_idiom_with_torch_optimizer():
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny()
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=True, num_bins=dims.tok.num_bins, model_name=None)
    model = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32)
    model.load_state_dict(synth.synth_state_dict(dims, 0))
    optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-3)
    tokens, attention_masks, Lp = synth.synth_batch(dims, 4, text_len=8, num_steps=4, max_traj_token=40)
    pcrgbs = torch.stack([synth.synth_cloud(dims, i) for i in range(4)]).cuda()
    tokens, attention_masks = tokens.cuda(), attention_masks.cuda()
    prompts = tokens[:, :Lp]
    model.train()
    losses = []
    torch.manual_seed(0)                                            # FPS start comes from the global RNG, as in the reference
    for _ in range(5):
        optimizer.zero_grad()
        outputs = model(input_ids=tokens, attention_mask=attention_masks, point_clouds=pcrgbs, return_dict=True)
        logits = outputs.logits[:, prompts.shape[1] - 1:-1, :]
        tgt = tokens[:, prompts.shape[1]:]
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.flatten(), ignore_index=dims.tok.pad)
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.3, losses
    frozen = [n for n, p in model.named_parameters() if n.startswith("model.point_backbone.")]
    assert all(dict(model.named_parameters())[n].grad is None for n in frozen)
    # two backward passes without zero_grad accumulate, like autograd
    optimizer.zero_grad()
    out = model(input_ids=tokens, attention_mask=attention_masks, point_clouds=pcrgbs, fps_start=[0, 0, 0, 0])
    l1 = out.logits.float().pow(2).mean()
    l1.backward()
    g1 = model.lm_head.weight.grad.clone()
    out = model(input_ids=tokens, attention_mask=attention_masks, point_clouds=pcrgbs, fps_start=[0, 0, 0, 0])
    out.logits.float().pow(2).mean().backward()
    assert torch.allclose(model.lm_head.weight.grad, 2 * g1, rtol=1e-4, atol=1e-7)
