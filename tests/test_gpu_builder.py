"""GPU: build_model(args) from a LOCAL HF-style directory (config.json + model.safetensors + tokenizer
files), the way train.py:67 / evaluate.py:79 call it: point-token ids wired from the tokenizer
(pointllm.py:277-300), trajectory tokens appended and embeddings grown without mean-init
(builder.py:33-46), save_pretrained / from-directory round trip, checkpoint dict load."""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu


def _make_tokenizer(path, vocab_size):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, vocab_size):
        vocab[f"w{i}"] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", pad_token="<pad>")
    fast.save_pretrained(path)


def test_build_model_from_local_directory(tmp_path):
    from egoscaler_amd.pointllm import build_model, PointLLMConfig, TrajPointLLMForCausalLM
    num_bins = 16
    dims = dims_tiny()
    base_vocab = dims.tok.point_patch                       # vocabulary before any added token
    dims.lm.vocab_size = base_vocab
    d = str(tmp_path / "PointLLM_tiny")
    cfg = PointLLMConfig.from_dims(dims)
    # a "pretrained PointLLM" directory: weights for the base vocabulary + 3 point tokens (upstream PointLLM layout)
    dims_pt = dims_tiny()
    dims_pt.lm.vocab_size = base_vocab + 3
    cfg.vocab_size = base_vocab + 3
    args0 = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=num_bins, model_name=None)
    m0 = TrajPointLLMForCausalLM(args0, dims_pt, None, device="cuda", dtype=torch.float32)
    m0.config = cfg
    m0.load_state_dict(synth.synth_state_dict(dims_pt, 0))
    m0.save_pretrained(d)
    _make_tokenizer(d, base_vocab)
    assert os.path.exists(os.path.join(d, "config.json")) and os.path.exists(os.path.join(d, "model.safetensors"))

    args = types.SimpleNamespace(model_name=d, num_bins=num_bins, unfreeze_pc_encoder=False, unfreeze_language_model=False)
    model, tokenizer, pbc, use_se = build_model(args)
    assert use_se is True
    # point tokens were appended to the tokenizer and their ids wired into the config dict and the kernels
    assert pbc["point_patch_token"] == base_vocab and pbc["point_start_token"] == base_vocab + 1 and pbc["point_end_token"] == base_vocab + 2
    assert {"point_cloud_dim", "backbone_output_dim", "project_output_dim", "point_token_len", "mm_use_point_start_end",
            "projection_hidden_layer", "projection_hidden_dim", "use_max_pool", "default_point_patch_token"} <= set(pbc)
    t = model.dims.tok
    assert (t.point_patch, t.point_start, t.point_end) == (base_vocab, base_vocab + 1, base_vocab + 2)
    # <ts> <tsep> <te> + num_bins <p*> tokens grow the vocabulary; old rows are preserved, new rows ~N(0, 0.02)
    V = base_vocab + 3 + 3 + num_bins
    assert len(tokenizer) == V and model.state_dict()["model.embed_tokens.weight"].shape[0] == V == model.state_dict()["lm_head.weight"].shape[0]
    assert (t.ts, t.tsep, t.te, t.p0) == tuple(tokenizer.convert_tokens_to_ids(["<ts>", "<tsep>", "<te>", "<p0>"]))
    old = synth.synth_state_dict(dims_pt, 0)["model.embed_tokens.weight"]
    assert torch.equal(model.state_dict()["model.embed_tokens.weight"][:base_vocab + 3].cpu(), old)
    new = model.state_dict()["model.embed_tokens.weight"][base_vocab + 3:]
    assert 0.005 < float(new.std()) < 0.04
    # the built model runs: tokens assembled with the tokenizer's ids
    dims_run = model.dims
    toks, masks, Lp = synth.synth_batch(dims_run, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims_run, i) for i in range(2)])
    model.train()
    loss = model.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, tokenizer.pad_token_id, fps_start=[0, 3])
    assert np.isfinite(float(loss))
    # reference checkpoint dict (train.py:287-296) loads back
    ck = {"epoch": 0, "model_state_dict": {k: v.cpu() for k, v in model.state_dict().items()}, "global_step": 1}
    torch.save(ck, tmp_path / "latest_model.pt")
    model2, _, _, _ = build_model(args)
    model2.load_state_dict(torch.load(tmp_path / "latest_model.pt", map_location="cpu", weights_only=True)["model_state_dict"])
    with torch.no_grad():
        a = model(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts.cuda(), fps_start=[0, 3]).logits
        b = model2(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts.cuda(), fps_start=[0, 3]).logits
    assert torch.equal(a, b)


def test_list_of_variable_size_clouds_matches_batched():
    """pointllm.py:117-122: point_clouds may be a list of [N_i, 6] tensors."""
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny()
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32).eval()
    m.load_state_dict(synth.synth_state_dict(dims, 0))
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)]).cuda()
    with torch.no_grad():
        a = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts, fps_start=[0, 17]).logits
        b = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=[pts[0], pts[1]], fps_start=[0, 17]).logits
        big = torch.cat([pts[1], pts[1][:100] + 5.0], 0)           # a larger cloud in slot 1 changes only sample 1
        c = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=[pts[0], big], fps_start=[0, 17]).logits
    assert torch.equal(a, b)
    assert torch.equal(a[0], c[0]) and not torch.equal(a[1], c[1])


def test_directory_written_by_the_reference_classes_loads(golden_dir, tmp_path):
    """VERDICT r2 missing #4: tests/golden/tiny_hf_dir is the directory the REFERENCE's own `save_pretrained` wrote (config.json as HF's
    PretrainedConfig serialises PointLLMConfig, generation_config.json, model.safetensors with HF's metadata header; oracle/gen_golden.py::
    gen_train_steps).  `PointLLMConfig.from_pretrained` + `TrajPointLLMForCausalLM(args, config, dir)` (model_arch.py:13-31) read it back: every key,
    every value, and the model computes the reference's logits."""
    import json
    import shutil
    from egoscaler_amd.pointllm import PointLLMConfig, TrajPointLLMForCausalLM
    src = os.path.join(golden_dir, "tiny_hf_dir")
    d = str(tmp_path / "ref_written")
    shutil.copytree(src, d)
    t = dims_tiny()
    with open(os.path.join(d, "tiny.yaml"), "w") as f:          # the name in config.json resolves to a YAML (pointllm.py:38-41)
        f.write("model : {\n  NAME: PointTransformer,\n  trans_dim: %d,\n  depth: %d,\n  drop_path_rate: 0.0,\n  cls_dim: 40,\n  num_heads: %d,\n  group_size: %d,\n"
                "  num_group: %d,\n  encoder_dims: %d,\n  point_dims: 3,\n  projection_hidden_layer: 2,\n  projection_hidden_dim: [%d, %d],\n  use_max_pool: false\n}\nnpoints: %d\n"
                % (t.pb.trans_dim, t.pb.depth, t.pb.num_heads, t.pb.group_size, t.pb.num_group, t.pb.encoder_dims, *t.pb.projection_hidden_dim, t.pb.npoints))
    cfg = PointLLMConfig.from_pretrained(d)
    gen = json.load(open(os.path.join(d, "generation_config.json")))
    assert (gen["eos_token_id"], gen["pad_token_id"]) == (cfg.eos_token_id, cfg.pad_token_id) == (2, 0)     # what generate() defaults to
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=t.tok.num_bins, model_name=d)
    m = TrajPointLLMForCausalLM(args, cfg, d, device="cuda", dtype=torch.float32).eval()
    sd, want = m.state_dict(), synth.synth_state_dict(t, 0)
    assert list(sd.keys()) == list(want.keys())
    for k, v in want.items():
        assert torch.equal(sd[k].cpu(), v), k
    m.set_point_token_ids(t.tok.point_patch, t.tok.point_start, t.tok.point_end)
    g = np.load(os.path.join(golden_dir, "tiny_model.npz"), allow_pickle=False)
    toks, masks, Lp = synth.synth_batch(t, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(t, i) for i in range(2)])
    with torch.no_grad():
        lg = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts.cuda(), fps_start=g["fps_start"]).logits
    assert float(np.abs(lg.float().cpu().numpy() - g["logits"]).max()) < 1e-3 * float(np.abs(g["logits"]).max())


def test_load_point_backbone_checkpoint(tmp_path):
    """PointTransformer.load_checkpoint (point_encoder.py:144-166, reached through pointllm.py:86-87): a `.pt` whose `state_dict` keys carry the
    `module.point_encoder.` prefix; other keys of the file are ignored, missing / unexpected encoder keys are reported, the encoder's output
    follows the loaded values (VERDICT r3 missing #4)."""
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny()
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32).eval()
    m.load_state_dict(synth.synth_state_dict(dims, 0))
    other = synth.synth_state_dict(dims, 5)                       # a different "pre-trained PointBERT"
    pre = "model.point_backbone."
    enc = {k[len(pre):]: v for k, v in other.items() if k.startswith(pre)}
    sd = {"module.point_encoder." + k: v for k, v in enc.items()}
    sd["module.cls_head_finetune.0.weight"] = torch.zeros(4, 4)   # not the encoder's: ignored
    sd["module.point_encoder.extra_token"] = torch.zeros(3)        # the encoder's, but unknown here: reported
    dropped = "blocks.blocks.0.attn.proj.bias"
    del sd["module.point_encoder." + dropped]                      # absent from the file: reported, the model keeps its value
    path = str(tmp_path / "pointbert.pt")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)]).cuda()
    toks, masks, _ = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    with torch.no_grad():
        before = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts, fps_start=[0, 3]).logits.clone()
    kept = m.state_dict()[pre + dropped].clone()
    res = m.get_model().load_point_backbone_checkpoint(path) if hasattr(m.get_model(), "load_point_backbone_checkpoint") else m.load_point_backbone_checkpoint(path)
    assert res.missing_keys == [dropped] and res.unexpected_keys == ["extra_token"]
    now = m.state_dict()
    for k, v in enc.items():
        assert torch.equal(now[pre + k].cpu(), kept.cpu() if k == dropped else v), k
    assert torch.equal(now["model.embed_tokens.weight"].cpu(), synth.synth_state_dict(dims, 0)["model.embed_tokens.weight"])
    with torch.no_grad():
        after = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pts, fps_start=[0, 3]).logits
    assert not torch.equal(before, after)                          # the folded BatchNorm / derived copies were rebuilt from the new values
    bad = dict(sd)
    bad["module.point_encoder.cls_token"] = torch.zeros(1, 1, 7)
    torch.save({"state_dict": bad}, path)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_point_backbone_checkpoint(path)
