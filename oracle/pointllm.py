"""Oracle for SURVEY.md §8a rows A9, A10, A12, A13 glue: projector, point-token splice, whole-model
forward, loss and greedy generation.  Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import llama as L
from . import pointbert as PB


def point_proj(sd, x, n_hidden):
    """Linear+GELU per hidden layer, final Linear (pointllm/model/pointllm.py:67-81)."""
    for j in range(n_hidden):
        x = F.gelu(F.linear(x, sd[f"model.point_proj.{2 * j}.weight"], sd[f"model.point_proj.{2 * j}.bias"]))
    j = n_hidden
    return F.linear(x, sd[f"model.point_proj.{2 * j}.weight"], sd[f"model.point_proj.{2 * j}.bias"])


def splice_positions(input_ids: torch.Tensor, tok, P: int):
    """Integer logic of pointllm.py:131-171 for mm_use_point_start_end=True: per sample the list of
    <point_start> positions; raises ValueError exactly where the reference does."""
    out = []
    for ids in input_ids:
        if (ids == tok.point_patch).sum() == 0:                    # :137 text-only sample
            out.append([])
            continue
        if (ids == tok.point_start).sum() != (ids == tok.point_end).sum():     # :146
            raise ValueError("The number of point start tokens and point end tokens should be the same.")
        starts = torch.where(ids == tok.point_start)[0].tolist()
        for s in starts:
            if s + P + 1 >= ids.shape[0] or ids[s + P + 1] != tok.point_end:   # :150
                raise ValueError("The point end token should follow the point start token.")
        out.append(starts)
    return out


def splice(input_ids, inputs_embeds, point_features, tok, P):
    """pointllm.py:131-171 for mm_use_point_start_end=True, statement by statement — including what it does with several segments in one
    sample: the sample's features are fetched ONCE (`point_features[cur_point_idx]`, :143, IndexError past the clouds given — before the
    sample's token checks), every pass of the `for point_start_token_pos` loop rebuilds the row from the ORIGINAL embeddings (:155), so only the
    LAST segment is spliced, and cur_point_idx advances once per segment (:156); a sample without patch tokens advances it by one (:137-142)."""
    rows = []
    cur = 0
    for b in range(input_ids.shape[0]):
        ids, e = input_ids[b], inputs_embeds[b]
        if (ids == tok.point_patch).sum() == 0:                                # :137 text-only sample
            rows.append(e)
            cur += 1
            continue
        feats = point_features[cur]                                            # :143
        if (ids == tok.point_start).sum() != (ids == tok.point_end).sum():     # :146
            raise ValueError("The number of point start tokens and point end tokens should be the same.")
        new = None
        for s in torch.where(ids == tok.point_start)[0].tolist():
            if s + P + 1 >= ids.shape[0] or ids[s + P + 1] != tok.point_end:   # :150 (past the end the reference raises IndexError; not restated)
                raise ValueError("The point end token should follow the point start token.")
            new = torch.cat((e[:s + 1], feats, e[s + P + 1:]), dim=0)          # :155
            cur += 1                                                           # :156
        if new is None:
            raise NotImplementedError("patch tokens without a <point_start>: the reference appends a stale / unbound variable here (:157)")
        rows.append(new)
    return torch.stack(rows, 0)


def forward(sd, dims, input_ids, attention_mask, point_clouds, fps_start, kv_cache=None, taps=None, pc_train=False, pc_drop=None):
    """Whole model -> logits [B,S,V] (pointllm.py:90-178,215-228).  pc_train=False: frozen eval-mode point
    backbone (the default flags); True: --unfreeze_pc_encoder in train() (gradients flow into it, BatchNorm
    uses batch statistics and updates the running stats held in `sd`)."""
    lm, pb, tok = dims.lm, dims.pb, dims.tok
    emb = F.embedding(input_ids, sd["model.embed_tokens.weight"])
    if point_clouds is not None and (input_ids.shape[1] != 1):
        if pc_train:
            feats = PB.point_transformer(sd, "model.point_backbone.", point_clouds, pb, fps_start, taps, training=True, drop=pc_drop)
        else:
            with torch.no_grad():
                feats = PB.point_transformer(sd, "model.point_backbone.", point_clouds, pb, fps_start, taps)
        feats = point_proj(sd, feats, len(pb.projection_hidden_dim))
        if taps is not None:
            taps["point_features"] = feats
        emb = splice(input_ids, emb, feats, tok, pb.point_token_len)
    if taps is not None:
        taps["inputs_embeds"] = emb
    h = L.decoder_stack(sd, emb, attention_mask, lm, kv_cache, taps)
    if taps is not None:
        taps["hidden"] = h
    return F.linear(h, sd["lm_head.weight"])


def greedy_generate(sd, dims, prompts, prompt_masks, point_clouds, fps_start, max_new_tokens):
    """model_arch.py:77-108 with do_sample=False: prefill (encoder + splice), then one token per
    step against the KV cache (pointllm.py:112,255-275).  Returns (sequences, scores list)."""
    lm = dims.lm
    cache = [dict() for _ in range(lm.num_hidden_layers)]
    ids = prompts
    mask = prompt_masks
    logits = forward(sd, dims, ids, mask, point_clouds, fps_start, cache)
    scores = []
    seq = ids
    for t in range(max_new_tokens):
        s = logits[:, -1, :].float()
        scores.append(s)
        nxt = s.argmax(-1, keepdim=True)
        seq = torch.cat([seq, nxt], dim=1)
        mask = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
        if t + 1 < max_new_tokens:
            logits = forward(sd, dims, nxt, mask, None, None, cache)
    return seq, scores
