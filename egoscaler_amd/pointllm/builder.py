"""build_model(args) -> (model, tokenizer, point_backbone_config, mm_use_point_start_end)

Mirrors egoscaler/models/pointllm/builder.py:9-55.  `args` needs .model_name, .num_bins,
.unfreeze_pc_encoder, .unfreeze_language_model (optionally .dtype / .device).  The tokenizer is the
HuggingFace one, exactly as in the reference (host-side string handling, not part of the GPU path);
model_name must be a local directory (there is no network)."""
import logging
import os

import torch

from .constant import RT2_TOKEN_TEMPLATE, TIMESTEP_START_TOKEN, TIMESTEP_SEP_TOKEN, TIMESTEP_END_TOKEN
from .model_arch import PointLLMConfig, TrajPointLLMForCausalLM


def init_model(args, tokenizer=None):
    model_name = os.path.expanduser(args.model_name)
    config = PointLLMConfig.from_pretrained(model_name)
    logging.warning(f"Model name: {os.path.basename(model_name)}")
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_name)
    model = TrajPointLLMForCausalLM(args, config, model_name, device=getattr(args, "device", "cuda"),
                                    dtype=getattr(args, "dtype", torch.float32))
    model.initialize_tokenizer_point_backbone_config_wo_embedding(tokenizer)
    mm_use_point_start_end = getattr(model.config, "mm_use_point_start_end", False)
    return model, tokenizer, model.get_model().point_backbone_config, mm_use_point_start_end


def add_trajectory_token(args, model, tokenizer):
    """builder.py:33-46: <ts> <tsep> <te> then <p0>..<p{num_bins-1}>; embeddings grow without mean-init."""
    if args.num_bins > 0:
        tokenizer.add_tokens([TIMESTEP_START_TOKEN, TIMESTEP_SEP_TOKEN, TIMESTEP_END_TOKEN])
        tokenizer.add_tokens([RT2_TOKEN_TEMPLATE.format(p=p) for p in range(args.num_bins)])
    model.resize_token_embeddings(len(tokenizer), mean_resizing=False)
    t = model.dims.tok
    ids = tokenizer.convert_tokens_to_ids([TIMESTEP_START_TOKEN, TIMESTEP_SEP_TOKEN, TIMESTEP_END_TOKEN, RT2_TOKEN_TEMPLATE.format(p=0)])
    if args.num_bins > 0:
        t.ts, t.tsep, t.te, t.p0, t.num_bins = ids[0], ids[1], ids[2], ids[3], args.num_bins
    return model, tokenizer


def build_model(args, tokenizer=None):
    model, tokenizer, point_backbone_config, mm_use_point_start_end = init_model(args, tokenizer)
    model, tokenizer = add_trajectory_token(args, model, tokenizer)
    return model, tokenizer, point_backbone_config, mm_use_point_start_end
