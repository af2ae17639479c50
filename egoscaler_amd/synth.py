"""Seeded synthetic weights, clips and token sequences.

No dataset, tokenizer file or checkpoint exists offline, and the reference's
`CustomDataset.__getitem__` is missing from the release, so the clip -> point
cloud -> token batch glue is defined here, following the fragments the reference does contain:
  * RGB-D un-projection order and masks   egoscaler/data/tools/pcm_tools.py:68-96
  * 8192-point clouds + pc_norm           pointbert/PointTransformer_8192point_2layer.yaml:16,
                                          pointllm/data/utils.py:146-157
  * token sequence layout                 pointllm/dataset.py:16-19,150-194, constant.py:11-25
  * 256-bin discretisation                pointllm/utils/utils.py:13-16

Everything is a pure function of (name|sample id, shape, seed) through numpy Philox, so the CPU
oracle, the GPU path, the golden fixtures and the bench all see identical bits.
"""
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import EgoDims, PointBertDims, LlamaDims

# Aria pin-hole camera constants (egoscaler/configs/camera.py:7-9, configs/data.py:3)
ARIA_IMAGE_SIZE = 1408
ARIA_FOCAL = 605.343
ARIA_PP = 703.5
DEPTH_THRESHOLD = 5.0


def _rng(seed: int, *keys: int) -> np.random.Generator:
    k = [int(seed) & 0xFFFFFFFFFFFFFFFF]
    acc = 0
    for i, x in enumerate(keys):
        acc = (acc * 0x9E3779B97F4A7C15 + (int(x) + 1) * (i + 1)) & 0xFFFFFFFFFFFFFFFF
    k.append(acc)
    return np.random.Generator(np.random.Philox(key=np.array(k, dtype=np.uint64)))


# --------------------------------------------------------------------------------------------
# state-dict layout (key names/shapes as the reference classes produce them)
# --------------------------------------------------------------------------------------------
def pointbert_param_shapes(pb: PointBertDims) -> List[Tuple[str, Tuple[int, ...]]]:
    D, C = pb.trans_dim, pb.encoder_dims
    out = [("cls_token", (1, 1, D)), ("cls_pos", (1, 1, D))]
    out += [("encoder.first_conv.0.weight", (pb.pn_c1, pb.point_dims, 1)), ("encoder.first_conv.0.bias", (pb.pn_c1,))]
    out += [(f"encoder.first_conv.1.{k}", (pb.pn_c1,)) for k in ("weight", "bias", "running_mean", "running_var")]
    out += [("encoder.first_conv.1.num_batches_tracked", ())]
    out += [("encoder.first_conv.3.weight", (pb.pn_c2, pb.pn_c1, 1)), ("encoder.first_conv.3.bias", (pb.pn_c2,))]
    out += [("encoder.second_conv.0.weight", (pb.pn_c3, 2 * pb.pn_c2, 1)), ("encoder.second_conv.0.bias", (pb.pn_c3,))]
    out += [(f"encoder.second_conv.1.{k}", (pb.pn_c3,)) for k in ("weight", "bias", "running_mean", "running_var")]
    out += [("encoder.second_conv.1.num_batches_tracked", ())]
    out += [("encoder.second_conv.3.weight", (C, pb.pn_c3, 1)), ("encoder.second_conv.3.bias", (C,))]
    out += [("reduce_dim.weight", (D, C)), ("reduce_dim.bias", (D,))]
    out += [("pos_embed.0.weight", (pb.pos_hidden, 3)), ("pos_embed.0.bias", (pb.pos_hidden,)),
            ("pos_embed.2.weight", (D, pb.pos_hidden)), ("pos_embed.2.bias", (D,))]
    for i in range(pb.depth):
        p = f"blocks.blocks.{i}."
        out += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
                (p + "mlp.fc1.weight", (pb.mlp_ratio * D, D)), (p + "mlp.fc1.bias", (pb.mlp_ratio * D,)),
                (p + "mlp.fc2.weight", (D, pb.mlp_ratio * D)), (p + "mlp.fc2.bias", (D,)),
                (p + "attn.qkv.weight", (3 * D, D)),
                (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    return out


def param_shapes(dims: EgoDims) -> List[Tuple[str, Tuple[int, ...]]]:
    lm, pb = dims.lm, dims.pb
    d, f, V = lm.hidden_size, lm.intermediate_size, lm.vocab_size
    out = [("model.embed_tokens.weight", (V, d))]
    for i in range(lm.num_hidden_layers):
        p = f"model.layers.{i}."
        out += [(p + f"self_attn.{n}_proj.weight", (d, d)) for n in "qkvo"]
        out += [(p + "mlp.gate_proj.weight", (f, d)), (p + "mlp.up_proj.weight", (f, d)),
                (p + "mlp.down_proj.weight", (d, f)),
                (p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,))]
    out += [("model.norm.weight", (d,))]
    out += [("model.point_backbone." + k, s) for k, s in pointbert_param_shapes(pb)]
    last = pb.trans_dim
    for j, h in enumerate(list(pb.projection_hidden_dim) + [d]):
        out += [(f"model.point_proj.{2 * j}.weight", (h, last)), (f"model.point_proj.{2 * j}.bias", (h,))]
        last = h
    out += [("lm_head.weight", (V, d))]
    return out


def synth_tensor(name: str, shape, seed: int = 0, std: float = 0.02) -> torch.Tensor:
    """Deterministic fp32 tensor for a state-dict entry, by name class."""
    g = _rng(seed, zlib.crc32(name.encode()))
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_var":
        a = g.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = g.normal(0.0, 0.1, size=shape)
    elif leaf == "weight" and len(shape) == 1:          # norm scales: around 1
        a = 1.0 + g.normal(0.0, 0.05, size=shape)
    elif leaf == "bias":
        a = g.normal(0.0, 0.02, size=shape)
    elif leaf in ("cls_token", "cls_pos"):
        a = g.normal(0.0, 0.5, size=shape)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        # keep activations O(1) through deep stacks at every width (std 0.02 at d=4096 ~ 1/sqrt(d)*1.3)
        s = std if fan_in >= 1024 else min(0.35, 1.0 / np.sqrt(fan_in))
        a = g.normal(0.0, s, size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(dims: EgoDims, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, s, seed) for k, s in param_shapes(dims)}


# --------------------------------------------------------------------------------------------
# synthetic clip -> RGB-D frames
# --------------------------------------------------------------------------------------------
def clip_intrinsics(H: int) -> Tuple[float, float]:
    """Aria intrinsics scaled to an HxH frame: f = 605.343*H/1408, pp = (H-1)/2."""
    return ARIA_FOCAL * H / ARIA_IMAGE_SIZE, (H - 1) / 2.0


def synth_clip(sample_id: int, T: int, H: int, W: int, seed: int = 42):
    """rgb u8 [T,H,W,3] uniform 1..255 with 5 % of pixels zeroed; depth f32 [T,H,W] in [0.3, 6.0)."""
    g = _rng(seed + sample_id, 0xC11F)
    rgb = g.integers(1, 256, size=(T, H, W, 3), dtype=np.uint8)
    zero = g.random(size=(T, H, W)) < 0.05
    rgb[zero] = 0
    depth = g.uniform(0.3, 6.0, size=(T, H, W)).astype(np.float32)
    return rgb, depth


# --------------------------------------------------------------------------------------------
# synthetic token sequence
# --------------------------------------------------------------------------------------------
def discretize(values: np.ndarray, num_bins: int) -> np.ndarray:
    """np.digitize(v, linspace(-1,1,num_bins)) - 1  (pointllm/utils/utils.py:13-16)."""
    return np.digitize(values, np.linspace(-1, 1, num_bins)) - 1


def synth_tokens(dims: EgoDims, sample_id: int, text_len: int = 16, num_steps: int = 20,
                 max_traj_token: int = 160, seed: int = 42):
    """[bos] desc_a <point_start> <point_patch>xP <point_end> desc_b <ts> (p*6 <tsep>)*steps <te> eos pad..

    Returns (tokens i64 [S], attention_mask bool [S], prompt_len) where prompt_len follows
    dataset.py:180-182: the prompt runs up to and including the first <tsep>.
    """
    t = dims.tok
    P = dims.pb.point_token_len
    g = _rng(seed + sample_id, 0x70C5)
    lo_vocab = min(t.point_patch, dims.lm.vocab_size)
    desc = g.integers(3, lo_vocab, size=text_len, dtype=np.int64)
    a, b = desc[: text_len // 2], desc[text_len // 2:]
    traj = g.uniform(-1, 1, size=(num_steps, 6))
    bins = np.clip(discretize(traj, t.num_bins), 0, t.num_bins - 1)
    tt = [t.ts]
    for s in range(num_steps):
        tt += [t.p0 + int(v) for v in bins[s]] + [t.tsep]
    tt += [t.te, t.eos]
    n_real = len(tt)
    if n_real > max_traj_token:
        raise ValueError("trajectory tokens exceed max_traj_token")
    tt += [t.pad] * (max_traj_token - n_real)
    head = [t.bos] + a.tolist() + [t.point_start] + [t.point_patch] * P + [t.point_end] + b.tolist()
    toks = np.array(head + tt, dtype=np.int64)
    mask = np.ones_like(toks, dtype=bool)
    mask[len(head) + n_real:] = False
    prompt_len = len(head) + 1 + 6 + 1          # <ts> + first step + first <tsep>
    return toks, mask, prompt_len


def synth_batch(dims: EgoDims, B: int, text_len: int = 16, num_steps: int = 20,
                max_traj_token: int = 160, seed: int = 42, first_id: int = 0):
    toks, masks = [], []
    pl = None
    for i in range(B):
        t, m, pl = synth_tokens(dims, first_id + i, text_len, num_steps, max_traj_token, seed)
        toks.append(t)
        masks.append(m)
    return torch.from_numpy(np.stack(toks)), torch.from_numpy(np.stack(masks)), pl


def synth_cloud(dims: EgoDims, sample_id: int, seed: int = 42) -> torch.Tensor:
    """A ready [N,6] f32 cloud (xyz in the unit ball after pc_norm-like scaling, rgb in [0,1]) for
    tests that start at the PointBERT input instead of at RGB-D frames."""
    g = _rng(seed + sample_id, 0xC10D)
    N = dims.pb.npoints
    xyz = g.normal(0, 1, size=(N, 3))
    xyz -= xyz.mean(0)
    xyz /= np.sqrt((xyz ** 2).sum(1)).max()
    rgb = g.uniform(0, 1, size=(N, 3))
    return torch.from_numpy(np.concatenate([xyz, rgb], 1).astype(np.float32))
