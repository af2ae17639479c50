"""GPU: BASELINE.json configs[1] at FULL size (32 LLaMA-7B layers, bs 8, S = 692, bf16, frozen-LLM mode) — no oracle finishes
this in seconds, so the checks are size-independent properties of the training step (the 2-layer slice of the same width is
compared with the oracle in tests/test_gpu_parity_7b.py):
  * additivity over the batch: the step on 8 clips == the token-weighted mean of the steps on its two halves (M = 5536 rows vs
    2 x 2768: different tile plans, tail plans and attention grids must agree), loss and gradients;
  * permutation of the samples permutes nothing in the loss and the summed gradients;
  * determinism: the same step twice gives bit-identical loss and gradients (no atomic reductions since round 4)."""
import math
import types

import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b

pytestmark = pytest.mark.gpu


def _fro(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.timeout(900)
def test_fullsize_step_additivity_permutation_determinism():
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_7b()
    dev = torch.device("cuda")
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=256, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(99)
    with torch.no_grad():
        for n, p in list(m.named_parameters()) + list(m.named_buffers()):
            leaf = n.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                continue
            if leaf == "running_var" or (leaf == "weight" and p.dim() == 1):
                p.fill_(1.0)
            elif leaf == "running_mean":
                p.zero_()
            else:
                fan = p[0].numel() if p.dim() > 1 else p.numel()
                for r0 in range(0, p.shape[0], 8192):
                    blk = p[r0:r0 + 8192]
                    blk.copy_(torch.empty(blk.shape, dtype=torch.float32, device=dev).normal_(0, 0.02 if fan >= 1024 else min(0.35, fan ** -0.5), generator=g))
    m.engine.prepared = False
    m.train()
    B = 8
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=16, num_steps=20, max_traj_token=160)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)]).to(dev)
    toks, masks = toks.to(dev), masks.to(dev)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    watch = ["lm_head.weight", "model.point_proj.4.weight", "model.point_proj.0.weight", "model.norm.weight"]

    def step(idx):
        idx = torch.as_tensor(idx, device=dev)
        loss = m.loss_and_backward(toks[idx], masks[idx], pts[idx], Lp, dims.tok.pad, fps_start=start[: len(idx)])
        return float(loss), {n: m.engine.main_grad[n].clone() for n in watch}, m.engine.main_grad["model.embed_tokens.weight"].clone()

    l8, g8, e8 = step(range(8))
    assert math.isfinite(l8) and 0.0 < l8 < 2.0 * math.log(dims.lm.vocab_size), l8
    # determinism (run twice, bit-equality).  Round 4: the loss (per-row values + one ordered sum), the norm-weight gradient
    # (owner blocks per column strip), the embedding gradient (owner block per token id, rows in increasing order) and the bias column sums
    # no longer meet through fp32 atomics, so EVERY watched tensor and the loss repeat bit for bit
    l8b, g8b, e8b = step(range(8))
    same = {n: bool(torch.equal(g8[n], g8b[n])) for n in watch}
    assert l8b == l8, (l8, l8b)
    assert all(same.values()), same
    assert torch.equal(e8b, e8)
    # additivity: equal token counts per sample, so the batch loss / gradient is the plain mean of the halves
    la, ga, _ = step(range(0, 4))
    lb, gb, _ = step(range(4, 8))
    assert abs(0.5 * (la + lb) - l8) < 2e-3 * abs(l8), (la, lb, l8)
    for n in watch:
        half = 0.5 * (ga[n] + gb[n])
        assert _fro(half, g8[n]) < 3e-2, (n, _fro(half, g8[n]))
    # permutation
    lp, gp, _ = step([5, 2, 7, 0, 3, 6, 1, 4])
    assert abs(lp - l8) < 1e-3 * abs(l8)
    for n in watch:
        assert _fro(gp[n], g8[n]) < 3e-2, (n, _fro(gp[n], g8[n]))
    print(f"[fullsize] loss {l8:.5f} (ln V = {math.log(dims.lm.vocab_size):.5f}); halves {la:.5f} {lb:.5f}; permuted {lp:.5f}; "
          + "; ".join(f"{n}: add {_fro(0.5 * (ga[n] + gb[n]), g8[n]):.1e} perm {_fro(gp[n], g8[n]):.1e}" for n in watch))
