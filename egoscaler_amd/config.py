"""Shape configuration of the EgoScaler trajectory generator (PointBERT -> projector -> LLaMA).

Values follow the reference's own config sources:
  * PointBERT v1.2 YAML  (pointllm/model/pointbert/PointTransformer_8192point_2layer.yaml:1-16)
  * point_backbone_config (pointllm/model/pointllm.py:49-59)
  * LLaMA-7B dims + vocabulary growth (pointllm/model/pointllm.py:233 "V(32003)", builder.py:33-46)
Only numbers are taken from there; the classes are this build's own.
"""
from dataclasses import dataclass, field, asdict
from typing import List


@dataclass
class PointBertDims:
    trans_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    group_size: int = 32      # M neighbours per group
    num_group: int = 512      # G groups
    encoder_dims: int = 256
    point_dims: int = 6       # xyz + rgb when use_color (pointllm.py:42-43)
    projection_hidden_dim: List[int] = field(default_factory=lambda: [1024, 2048])
    npoints: int = 8192
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5
    drop_path_rate: float = 0.1   # stochastic depth of the ViT blocks, train mode only (YAML:5, point_encoder.py:133)
    # fixed by the reference's module definitions, not configurable there either:
    pos_hidden: int = 128     # pos_embed Linear(3,128) (point_encoder.py:127-131)
    mlp_ratio: int = 4        # Block(mlp_ratio=4.) (point_encoder.py:59)
    pn_c1: int = 128          # mini-PointNet widths (dvae.py:193-204)
    pn_c2: int = 256
    pn_c3: int = 512

    @property
    def head_dim(self):
        return self.trans_dim // self.num_heads

    @property
    def point_token_len(self):
        return self.num_group + 1   # cls + groups (pointllm.py:53)


@dataclass
class LlamaDims:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    vocab_size: int = 32003 + 3 + 256   # PointLLM vocab + <ts><tsep><te> + num_bins (builder.py:39-44)
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 2048

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


@dataclass
class SpecialTokens:
    """Token ids the splice and the data glue need (pointllm.py:277-300, builder.py:33-46)."""
    bos: int = 1
    eos: int = 2
    pad: int = 0
    point_patch: int = 32000
    point_start: int = 32001
    point_end: int = 32002
    ts: int = 32003
    tsep: int = 32004
    te: int = 32005
    p0: int = 32006           # <p0>; <p{i}> = p0 + i
    num_bins: int = 256


@dataclass
class EgoDims:
    pb: PointBertDims = field(default_factory=PointBertDims)
    lm: LlamaDims = field(default_factory=LlamaDims)
    tok: SpecialTokens = field(default_factory=SpecialTokens)

    def to_dict(self):
        return asdict(self)


def dims_7b() -> EgoDims:
    """PointLLM-7B v1.2 shapes: the configuration BASELINE.json's metric is quoted on."""
    return EgoDims()


def dims_tiny(vocab: int = 320, num_bins: int = 16) -> EgoDims:
    """Small LLaMA (d=128, L=2, H=4) around the full-size PointBERT front end is too slow on CPU,
    so the tiny config also shrinks PointBERT (but keeps every structural feature: cls token,
    pos re-added per block, 2-layer projector, colour channels)."""
    pb = PointBertDims(trans_dim=96, depth=2, num_heads=3, group_size=16, num_group=32,
                       encoder_dims=64, point_dims=6, projection_hidden_dim=[64, 96],
                       npoints=512, drop_path_rate=0.0)
    lm = LlamaDims(hidden_size=128, intermediate_size=352, num_hidden_layers=2,
                   num_attention_heads=4, vocab_size=vocab, max_position_embeddings=512)
    base = vocab - (3 + 3 + num_bins)
    tok = SpecialTokens(point_patch=base, point_start=base + 1, point_end=base + 2,
                        ts=base + 3, tsep=base + 4, te=base + 5, p0=base + 6, num_bins=num_bins)
    return EgoDims(pb=pb, lm=lm, tok=tok)
