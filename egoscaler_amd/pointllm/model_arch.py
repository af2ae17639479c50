"""TrajPointLLMForCausalLM on MI355X.

Mirrors egoscaler/models/pointllm/model_arch.py:8-124 (class, constructor arguments, freeze logic,
narrow forward, generate wrapper, train(mode) override) and the parts of
pointllm/model/pointllm.py it inherits (point_backbone_config :49-59,288-300; resize of the token
embeddings used by builder.py:44).  State-dict keys and shapes are the reference's,
so checkpoints round-trip.  All arithmetic runs in egoscaler_amd.engine on libegomi.so.
"""
import json
import os
import types
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ..config import EgoDims, LlamaDims, PointBertDims, SpecialTokens
from ..engine import Engine
from .. import ops, synth


class PointLLMConfig:
    """Duck-typed stand-in for the HF config the reference passes around (PointLLMConfig,
    pointllm.py:23-24): plain attributes, readable from / writable to a HF-style config.json."""
    model_type = "pointllm"

    def __init__(self, **kw):
        self.hidden_size = 4096
        self.intermediate_size = 11008
        self.num_hidden_layers = 32
        self.num_attention_heads = 32
        self.vocab_size = 32003
        self.rms_norm_eps = 1e-6
        self.rope_theta = 10000.0
        self.max_position_embeddings = 2048
        self.pad_token_id = 0
        self.bos_token_id = 1
        self.eos_token_id = 2
        self.point_backbone = "PointBERT"
        self.point_backbone_config_name = "PointTransformer_8192point_2layer"
        self.use_color = True
        self.mm_use_point_start_end = True
        self.DEFAULT_POINT_PATCH_TOKEN = "<point_patch>"
        self.DEFAULT_POINT_START_TOKEN = "<point_start>"
        self.DEFAULT_POINT_END_TOKEN = "<point_end>"
        self.point_bert = None            # optional dict overriding the PointBERT YAML values
        self.__dict__.update(kw)

    # PointBERT shapes by `point_backbone_config_name` (pointllm.py:38-41 resolves the name to a YAML next to its own sources; the numbers
    # of the two YAMLs the reference ships: pointbert/PointTransformer_8192point_2layer.yaml:1-16, PointTransformer_base_8192point.yaml:1-13)
    POINTBERT_BY_NAME = {
        "PointTransformer_8192point_2layer": dict(trans_dim=384, depth=12, num_heads=6, group_size=32, num_group=512, encoder_dims=256,
                                                  projection_hidden_dim=[1024, 2048], npoints=8192, drop_path_rate=0.1),
        "PointTransformer_base_8192point": dict(trans_dim=1152, depth=12, num_heads=12, group_size=48, num_group=512, encoder_dims=512,
                                                projection_hidden_dim=[], npoints=8192, drop_path_rate=0.1),
    }

    @classmethod
    def from_pretrained(cls, path, **overrides):
        """Reads the config.json of an HF directory — written by this class OR by the reference's own `save_pretrained` (HF PretrainedConfig:
        extra fields such as `architectures`, `dtype`, `transformers_version`, `head_dim`, `hidden_act` are kept as plain attributes;
        transformers >= 5 nests rope_theta under `rope_parameters`, 4.x wrote it at top level).  `point_backbone_config_name` selects the
        PointBERT shapes like the reference's YAML lookup (pointllm.py:38-41); a `<name>.yaml` inside the directory, or `point_bert=...`
        here / in the JSON, overrides the two names the reference ships."""
        with open(os.path.join(path, "config.json")) as f:
            kw = json.load(f)
        rp = kw.get("rope_parameters")
        if isinstance(rp, dict) and "rope_theta" in rp and "rope_theta" not in kw:
            kw["rope_theta"] = float(rp["rope_theta"])
            if rp.get("rope_type", "default") != "default":
                raise NotImplementedError(f"rope_type {rp.get('rope_type')!r}: only the default rotary embedding is built")
        if kw.get("rope_scaling"):
            raise NotImplementedError("rope_scaling is not built (the reference's checkpoints do not use it)")
        nkv = kw.get("num_key_value_heads")
        if nkv is not None and nkv != kw.get("num_attention_heads", nkv):
            raise NotImplementedError("grouped-query attention (num_key_value_heads != num_attention_heads) is not built")
        if kw.get("hidden_act", "silu") != "silu" or kw.get("attention_bias") or kw.get("mlp_bias") or kw.get("tie_word_embeddings"):
            raise NotImplementedError("config asks for a LLaMA variant this build does not have (hidden_act / biases / tied embeddings)")
        kw.update(overrides)
        cfg = cls(**kw)
        if cfg.point_bert is None:
            name = cfg.point_backbone_config_name
            yml = os.path.join(path, f"{name}.yaml")
            if os.path.exists(yml):
                import yaml
                with open(yml) as f:
                    y = yaml.safe_load(f)
                mdl = y.get("model", {})
                keys = ("trans_dim", "depth", "num_heads", "group_size", "num_group", "encoder_dims", "drop_path_rate")
                cfg.point_bert = {k: mdl[k] for k in keys if k in mdl}
                cfg.point_bert["projection_hidden_dim"] = list(mdl.get("projection_hidden_dim", [])) if mdl.get("projection_hidden_layer", 0) else []
                if "npoints" in y:
                    cfg.point_bert["npoints"] = y["npoints"]
            elif name in cls.POINTBERT_BY_NAME:
                cfg.point_bert = dict(cls.POINTBERT_BY_NAME[name])
            else:
                raise ValueError(f"unknown point_backbone_config_name {name!r}: put {name}.yaml into {path} or pass point_bert=dict(...)")
        return cfg

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump({k: v for k, v in self.__dict__.items()}, f, indent=1)

    @classmethod
    def from_dims(cls, dims: EgoDims):
        pb = dims.pb
        return cls(hidden_size=dims.lm.hidden_size, intermediate_size=dims.lm.intermediate_size,
                   num_hidden_layers=dims.lm.num_hidden_layers, num_attention_heads=dims.lm.num_attention_heads,
                   vocab_size=dims.lm.vocab_size, rms_norm_eps=dims.lm.rms_norm_eps, rope_theta=dims.lm.rope_theta,
                   max_position_embeddings=dims.lm.max_position_embeddings, pad_token_id=dims.tok.pad,
                   bos_token_id=dims.tok.bos, eos_token_id=dims.tok.eos,
                   point_bert=dict(trans_dim=pb.trans_dim, depth=pb.depth, num_heads=pb.num_heads, group_size=pb.group_size,
                                   num_group=pb.num_group, encoder_dims=pb.encoder_dims,
                                   projection_hidden_dim=list(pb.projection_hidden_dim), npoints=pb.npoints))

    def to_dims(self) -> EgoDims:
        pbk = dict(self.point_bert or {})
        pb = PointBertDims(point_dims=6 if self.use_color else 3, **pbk)
        lm = LlamaDims(hidden_size=self.hidden_size, intermediate_size=self.intermediate_size,
                       num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                       vocab_size=self.vocab_size, rms_norm_eps=self.rms_norm_eps,
                       rope_theta=getattr(self, "rope_theta", 10000.0), max_position_embeddings=self.max_position_embeddings)
        return EgoDims(pb=pb, lm=lm, tok=SpecialTokens(pad=self.pad_token_id or 0, bos=self.bos_token_id, eos=self.eos_token_id))


@dataclass
class CausalLMOutput:
    """Same fields the reference reads from CausalLMOutputWithPast (train.py:174, evaluate.py)."""
    logits: torch.Tensor
    loss: Optional[torch.Tensor] = None
    past_key_values: Optional[object] = None
    hidden_states: Optional[object] = None
    attentions: Optional[object] = None

    def __getitem__(self, i):
        return (self.logits,)[i]


@dataclass
class GenerateOutput:
    sequences: torch.Tensor
    scores: Tuple[torch.Tensor, ...]


def _install(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


class _LogitsFn(torch.autograd.Function):
    """Bridges the engine into torch.autograd so `loss.backward()` of the reference's training loop
    (train.py:176-183) drives the hand-written backward."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, attention_mask, point_clouds, fps_start, save):
        eng = model.engine
        hn = eng.forward_hidden(input_ids, attention_mask, point_clouds, fps_start, save=save)
        logits = eng.logits(hn)
        ctx.model, ctx.hn, ctx.saved = model, hn, save
        B, S = input_ids.shape
        return logits.view(B, S, -1)

    @staticmethod
    def backward(ctx, d_logits):
        model = ctx.model
        if not ctx.saved:
            raise RuntimeError("backward through a forward that ran without grad")
        eng = model.engine
        model._begin_backward()
        dl = d_logits.reshape(-1, d_logits.shape[-1])
        if dl.dtype != eng.dtype or not dl.is_contiguous():
            dl = dl.to(eng.dtype).contiguous()
        d_hn = eng.backward_logits(dl, ctx.hn)
        eng.backward_hidden(d_hn)
        model._publish_grads()
        return (torch.zeros_like(model._anchor), None, None, None, None, None, None)


class TrajPointLLMForCausalLM(nn.Module):
    def __init__(self, args, config, model_name: Optional[str] = None, device=None, dtype=torch.float32):
        super().__init__()
        self.args, self.config, self.model_name = args, config, model_name
        self.dims = config.to_dims() if not isinstance(config, EgoDims) else config
        dev = torch.device(device or "cuda")
        if dev.type != "cuda":
            raise RuntimeError("TrajPointLLMForCausalLM (egoscaler_amd) runs on an MI355X only; there is no CPU path")
        # q|k|v and gate|up of a decoder layer live side by side in ONE allocation each: every Parameter keeps its own name, shape and
        # contiguous rows (state_dict / load_state_dict / optimizers see the reference's tensors), and the engine reads [Wq;Wk;Wv] and
        # [Wgate;Wup] as stacked operands without holding copies that would have to follow each optimizer step
        shapes = dict(synth.param_shapes(self.dims))
        shared = {}
        for l in range(self.dims.lm.num_hidden_layers):
            for grp in (tuple(f"model.layers.{l}.self_attn.{n}_proj.weight" for n in "qkv"),
                        tuple(f"model.layers.{l}.mlp.{n}_proj.weight" for n in ("gate", "up"))):
                if all(g in shapes and len(shapes[g]) == 2 and shapes[g][1] == shapes[grp[0]][1] for g in grp):
                    block = torch.zeros(sum(shapes[g][0] for g in grp), shapes[grp[0]][1], dtype=dtype, device=dev)
                    r = 0
                    for g in grp:
                        shared[g] = block[r:r + shapes[g][0]]
                        r += shapes[g][0]
        for k, shape in synth.param_shapes(self.dims):
            leaf = k.rsplit(".", 1)[-1]
            is_buf = leaf in _BUFFER_LEAVES
            t = shared[k] if k in shared else torch.zeros(shape, dtype=torch.long if leaf == "num_batches_tracked" else dtype, device=dev)
            _install(self, k, t, is_buf)
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self.training_graph = True
        self.accumulate_grads = False
        self._tensors = None
        self.engine = None
        self.point_backbone_config = None
        self._build_engine()
        if model_name is not None and os.path.isdir(model_name):
            self.load_pretrained_weights()
        self._configure_trainable_parameters()

    # -- plumbing ---------------------------------------------------------------------------------
    def _build_engine(self):
        tensors = {k: v for k, v in self.named_parameters()}
        tensors.update({k: v for k, v in self.named_buffers()})
        dtype = self.model.embed_tokens.weight.dtype
        dev = self.model.embed_tokens.weight.device
        old = self.engine
        self.engine = Engine(self.dims, {k: v.data for k, v in tensors.items()}, dev, dtype)
        self.engine.param_ref = dict(self.named_parameters())
        if old is not None:
            self.engine.trainable = old.trainable
        pb = self.dims.pb
        cfg = self.point_backbone_config or {}
        cfg.update({"point_cloud_dim": pb.point_dims, "backbone_output_dim": pb.trans_dim,
                    "project_output_dim": self.dims.lm.hidden_size, "point_token_len": pb.point_token_len,
                    "mm_use_point_start_end": True, "projection_hidden_layer": len(pb.projection_hidden_dim),
                    "projection_hidden_dim": list(pb.projection_hidden_dim), "use_max_pool": False})     # pointllm.py:49-59
        self.point_backbone_config = cfg
        self.model.point_backbone_config = cfg
        self.model.load_point_backbone_checkpoint = self.load_point_backbone_checkpoint     # the reference calls it on get_model() (pointllm.py:86)

    def get_model(self):
        return self.model

    def state_dict(self, *a, **k):
        if getattr(self, "engine", None) is not None:
            self.engine.wait_param_updates()                # an overlapped optimizer step (EgoAdamW.step(overlap=True)) may still be writing parameters
        return super().state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        if getattr(self, "engine", None) is not None:
            self.engine.wait_param_updates()
        r = super()._apply(fn, *a, **k)
        if getattr(self, "engine", None) is not None:
            self._build_engine()
        return r

    def load_state_dict(self, sd, strict=True, **kw):
        self.engine.wait_param_updates()
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.engine.prepared = False
        self.engine.lm_wT_stale = True          # the padded lm_head transpose follows the loaded values
        return r

    def load_pretrained_weights(self):
        """HF directory: *.safetensors shards or pytorch_model*.bin (model_arch.py:25-31)."""
        files = sorted(f for f in os.listdir(self.model_name) if f.endswith(".safetensors"))
        sd = {}
        if files:
            from safetensors.torch import load_file
            for f in files:
                sd.update(load_file(os.path.join(self.model_name, f)))
        else:
            for f in sorted(f for f in os.listdir(self.model_name) if f.startswith("pytorch_model") and f.endswith(".bin")):
                sd.update(torch.load(os.path.join(self.model_name, f), map_location="cpu", weights_only=True))
        if sd:
            self.load_state_dict(sd, strict=False)

    def load_point_backbone_checkpoint(self, checkpoint_path=None):
        """pointllm.py:86-87 -> PointTransformer.load_checkpoint (point_encoder.py:144-166): a PointBERT pre-training checkpoint, `torch.save`d as
        {'state_dict': {...}}; the keys prefixed `module.point_encoder.` are the encoder's own state dict and are loaded non-strictly, every
        other key of the file (classification heads, the dVAE) is ignored.  Returns the (missing_keys, unexpected_keys) the reference prints.
        The file is read with weights_only=True: nothing in it can execute."""
        path = checkpoint_path if checkpoint_path is not None else getattr(self.config, "point_backbone_ckpt", None)
        if path is None:
            raise ValueError("no checkpoint path: pass one or set config.point_backbone_ckpt")
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        pre, dst = "module.point_encoder.", "model.point_backbone."
        src = {k[len(pre):]: v for k, v in ckpt["state_dict"].items() if k.startswith(pre)}
        own = {k[len(dst):]: v for k, v in super().state_dict().items() if k.startswith(dst)}
        self.engine.wait_param_updates()
        with torch.no_grad():
            for k, v in src.items():
                if k in own:
                    if own[k].shape != v.shape:
                        raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                                           f"the shape in current model is {tuple(own[k].shape)}.")      # load_state_dict raises on shapes even when not strict
                    own[k].copy_(v)
        missing = [k for k in own if k not in src]
        unexpected = [k for k in src if k not in own]
        self.engine.prepared = False                        # BatchNorm fold / stacked copies follow the loaded values
        if not missing and not unexpected:
            print(f"PointBERT's weights are successfully loaded from {path}")
        else:
            print("missing_keys", missing, "unexpected_keys", unexpected)
        return types.SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        if isinstance(self.config, PointLLMConfig):
            self.config.save_pretrained(path)
        os.makedirs(path, exist_ok=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def _configure_trainable_parameters(self):
        """model_arch.py:33-51: point backbone frozen unless --unfreeze_pc_encoder; model.layers
        frozen unless --unfreeze_language_model; embed_tokens always trainable; everything else
        (point_proj, model.norm, lm_head) keeps requires_grad=True."""
        unfreeze_pc = bool(getattr(self.args, "unfreeze_pc_encoder", False))
        unfreeze_llm = bool(getattr(self.args, "unfreeze_language_model", False))
        names = []
        for n, p in self.named_parameters():
            if n.startswith("model.point_backbone."):
                p.requires_grad = unfreeze_pc
            elif n.startswith("model.layers."):
                p.requires_grad = unfreeze_llm
            else:
                p.requires_grad = True
            if p.requires_grad:
                names.append(n)
        self.engine.set_trainable(names)

    def resize_token_embeddings(self, new_num_tokens, mean_resizing=False):
        """builder.py:44: grow embed_tokens / lm_head; new rows N(0, 0.02) like HF with
        mean_resizing=False (std = initializer_range)."""
        old = self.dims.lm.vocab_size
        if new_num_tokens == old:
            return
        for name in ("model.embed_tokens.weight", "lm_head.weight"):
            mod, leaf = self, name
            parts = name.split(".")
            for q in parts[:-1]:
                mod = getattr(mod, q)
            w = getattr(mod, parts[-1])
            nw = torch.empty(new_num_tokens, w.shape[1], dtype=w.dtype, device=w.device)
            n = min(old, new_num_tokens)
            nw[:n] = w.data[:n]
            if new_num_tokens > old:
                nw[old:].normal_(0.0, 0.02)
            mod.register_parameter(parts[-1], nn.Parameter(nw, requires_grad=w.requires_grad))
        self.dims.lm.vocab_size = new_num_tokens
        if hasattr(self.config, "vocab_size"):
            self.config.vocab_size = new_num_tokens
        tr = self.engine.trainable
        self.engine = None
        self._build_engine()
        self.engine.trainable = tr
        self.engine.main_grad, self.engine.layer_flat = {}, {}

    def initialize_tokenizer_point_backbone_config_wo_embedding(self, tokenizer):
        """pointllm.py:277-300."""
        cfg = self.point_backbone_config
        pp = getattr(self.config, "DEFAULT_POINT_PATCH_TOKEN", "<point_patch>")
        tokenizer.add_tokens([pp], special_tokens=True)
        cfg["default_point_patch_token"] = pp
        cfg["point_patch_token"] = tokenizer.convert_tokens_to_ids([pp])[0]
        ps = getattr(self.config, "DEFAULT_POINT_START_TOKEN", "<point_start>")
        pe = getattr(self.config, "DEFAULT_POINT_END_TOKEN", "<point_end>")
        tokenizer.add_tokens([ps, pe], special_tokens=True)
        cfg["default_point_start_token"], cfg["default_point_end_token"] = ps, pe
        cfg["point_start_token"] = tokenizer.convert_tokens_to_ids([ps])[0]
        cfg["point_end_token"] = tokenizer.convert_tokens_to_ids([pe])[0]
        self.set_point_token_ids(cfg["point_patch_token"], cfg["point_start_token"], cfg["point_end_token"])

    def set_point_token_ids(self, patch, start, end):
        t = self.dims.tok
        t.point_patch, t.point_start, t.point_end = int(patch), int(start), int(end)
        self.point_backbone_config.update(point_patch_token=int(patch), point_start_token=int(start), point_end_token=int(end))

    # -- gradients ----------------------------------------------------------------------------------
    def _begin_backward(self, explicit=False):
        """explicit=True (loss_and_backward): gradients are overwritten unless `accumulate_grads` is set, for both dtypes.
        explicit=False (torch autograd driving `_LogitsFn`): fp32 follows torch semantics (`p.grad is None` after
        `optimizer.zero_grad()` means fresh, otherwise accumulate); bf16 has no `.grad` and follows `accumulate_grads`."""
        eng = self.engine
        eng._xt_last.clear()                 # transposed-activation cache of engine._wgrad is valid within one backward only
        for n, p in self.named_parameters():
            if n in eng.trainable:
                g = eng.grad_buffer(n)
                fresh = (p.grad is None) if (p.dtype == torch.float32 and not explicit) else (not self.accumulate_grads)
                if fresh:
                    if eng.lazy_zero_ok(n):
                        eng.grad_fresh.add(n)          # overwritten by its first wgrad product (engine._wgrad)
                    else:
                        g.zero_()

    def _publish_grads(self):
        eng = self.engine
        eng.flush_fresh()
        for n, p in self.named_parameters():
            if n in eng.trainable:
                g = eng.reduced_grad.get(n, eng.main_grad[n])     # (resident exchange: the bf16 view the optimizer will read)
                p.main_grad = g
                if p.dtype == torch.float32:
                    p.grad = g

    # -- reference API --------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, point_clouds=None, return_dict=None,
                fps_start=None):
        """model_arch.py:53-75: only input_ids / attention_mask / point_clouds / return_dict are used."""
        dev = self.engine.device
        input_ids = input_ids.to(dev)
        if fps_start is None and point_clouds is not None:
            n = point_clouds[0].shape[0] if isinstance(point_clouds, (list, tuple)) else point_clouds.shape[1]
            b = len(point_clouds) if isinstance(point_clouds, (list, tuple)) else point_clouds.shape[0]
            fps_start = torch.randint(0, n, (b,), dtype=torch.long)            # misc.py:52 (global CPU RNG)
        save = torch.is_grad_enabled() and self.training_graph
        logits = _LogitsFn.apply(self._anchor, self, input_ids, attention_mask, point_clouds, fps_start, save)
        out = CausalLMOutput(logits=logits)
        return out if (return_dict is None or return_dict) else (logits,)

    def loss_and_backward(self, input_ids, attention_mask, point_clouds, prompt_len, pad_token_id, fps_start=None,
                          backward=True, grad_scale=1.0):
        """Fused training step of train.py:166-183: forward, lm_head restricted to the trajectory span
        [Lp-1, S-1), cross-entropy with ignore_index=pad (mean), and the full backward, without
        materialising [B,S,V] logits.  Returns the loss (fp32 scalar tensor)."""
        eng = self.engine
        dev = eng.device
        input_ids = input_ids.to(dev)
        B, S = input_ids.shape
        if fps_start is None and point_clouds is not None:
            fps_start = torch.randint(0, point_clouds.shape[1], (B,), dtype=torch.long)
        hn = eng.forward_hidden(input_ids, attention_mask, point_clouds, fps_start, save=backward)
        d = hn.shape[1]
        Lp = int(prompt_len)
        hs = hn.view(B, S, d)[:, Lp - 1:S - 1].reshape(-1, d)
        tg = input_ids[:, Lp:].reshape(-1).contiguous()
        lg = eng.logits(hs, padded=backward)
        ls, cnt = ops.cross_entropy(lg, tg, pad_token_id, dlogits=lg if backward else None, grad_scale=grad_scale)
        loss = ls / cnt.float()
        if backward:
            self._begin_backward(explicit=True)
            d_hs = eng.backward_logits(lg, hs)
            d_hn = eng.ws.get("d_hn_full", (B, S, d), eng.dtype, zero=True)
            d_hn[:, Lp - 1:S - 1] = d_hs.view(B, S - Lp, d)
            eng.backward_hidden(d_hn.view(B * S, d))
            self._publish_grads()
        return loss[0]

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, point_clouds=None, max_length=20, temperature=1.0, top_k=50,
                 top_p=0.95, repetition_penalty=1.0, do_sample=True, num_return_sequences=1, fps_start=None,
                 eos_token_id="config", pad_token_id=None, seed=None, **kwargs):
        """model_arch.py:77-108: `max_length` means max_new_tokens; returns .sequences [B,S0+T'] and .scores (T' x [B,V], the PROCESSED
        scores, as HF returns them with output_scores=True).  Prefill runs encoder + splice and fills the KV cache; every later step
        feeds one token (the behaviour pointllm.py:112,255-275 intends; see DESIGN.md on the reference's cache bug).

        Every mode runs on the device under one hipGraph (decode.Decoder.sample -> egomi_sample_rows): repetition penalty, temperature,
        top-k, top-p in HF's order and with HF's tie rules (golden: tests/golden/sampling.npz, recorded from HF's own processors on the
        reference model's logits), a draw from the softmax (Gumbel-max on a counter-based generator seeded from torch's CPU generator —
        torch.multinomial's stream cannot be reproduced, so parity is on `.scores` and, with do_sample=False, on the ids), and HF's
        eos rule: `eos_token_id` defaults to the config's (GenerationConfig.from_model_config), finished rows emit `pad_token_id`, and
        the outputs are cut after the step at which every row has finished.  eos_token_id=None: fixed length, never stops."""
        from ..decode import Decoder
        eng = self.engine
        eng.wait_param_updates()                           # the decoder reads the weights outside the engine's forward pass
        dev = eng.device
        ids = input_ids.to(dev)
        if fps_start is None and point_clouds is not None:
            fps_start = torch.randint(0, point_clouds.shape[1], (ids.shape[0],), dtype=torch.long)
        n_ret = int(num_return_sequences)
        if n_ret > 1:                                      # HF expands every input n times (generation/utils.py _expand_inputs_for_generation)
            if isinstance(point_clouds, (list, tuple)):
                raise NotImplementedError("num_return_sequences > 1 with a list of ragged clouds is not built")
            ids = ids.repeat_interleave(n_ret, 0)
            attention_mask = None if attention_mask is None else attention_mask.to(dev).repeat_interleave(n_ret, 0)
            point_clouds = None if point_clouds is None else point_clouds.to(dev).repeat_interleave(n_ret, 0)
            fps_start = None if fps_start is None else torch.as_tensor(fps_start).to(dev).repeat_interleave(n_ret, 0)
        B, S0 = ids.shape
        T = int(max_length)
        if isinstance(eos_token_id, str):
            eos_token_id = self.dims.tok.eos
        if eos_token_id is not None and pad_token_id is None:
            pad_token_id = self.dims.tok.pad if self.dims.tok.pad is not None else eos_token_id     # HF: pad defaults to eos when the config has none
        for name, v in (("temperature", temperature), ("repetition_penalty", repetition_penalty)):
            if v is not None and not float(v) > 0:
                raise ValueError(f"`{name}` has to be a strictly positive float, but is {v}")               # logits_process.py:239,307
        if top_p is not None and not (0 < float(top_p) <= 1.0):
            raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")
        # one Decoder (static KV cache + captured token loops) per geometry, kept while the decoder layers it holds stacked copies of cannot
        # change: frozen-LLM mode, same prepared weights.  run_validation / evaluate (train.py:207-264, evaluate.py:104-154) call generate()
        # once per batch: without this every batch re-allocated the cache and re-captured a graph of (new tokens x ~300) kernels
        key = (B, S0 + T, eng.prepare_epoch)
        cache = self.__dict__.setdefault("_decoders", {})
        reuse = not eng.any_layer_trainable and os.environ.get("EGOMI_DECODER_CACHE", "1") != "0"
        dec = cache.get(key) if reuse else None
        if dec is None:
            dec = Decoder(eng, B, S0 + T)
            if reuse:
                while len(cache) >= 2:                     # the full batch and the split's short last one; a cache is 2 * L * B * H * Smax * hd elements
                    cache.pop(next(iter(cache)))
                cache[key] = dec
        dec.prefill(ids, attention_mask, point_clouds, fps_start, T)
        if not do_sample:                                  # HF applies the warpers (temperature / top-k / top-p) in sampling mode only
            temperature, top_k, top_p = 1.0, 0, 1.0
        seq, sc = dec.sample(T, do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                             eos=eos_token_id, pad=pad_token_id, seed=seed, use_graph=kwargs.get("use_graph", True))
        stop = T
        if eos_token_id is not None and T > 0:             # HF leaves the loop after the step at which the last unfinished row emitted eos
            hit = seq[:, S0:] == eos_token_id
            first = torch.where(hit.any(1), hit.int().argmax(1), torch.full((B,), T - 1, device=dev))
            stop = int(first.max()) + 1
        sc = sc[:stop].clone()                              # the decoder's buffers are static (and the decoder may be reused by the next call):
        return GenerateOutput(sequences=seq[:, :S0 + stop].clone(), scores=tuple(sc[t] for t in range(stop)))     # hand out copies

    def train(self, mode: bool = True):
        """model_arch.py:110-124: frozen parts stay in eval(); embed_tokens follows `mode`."""
        super().train(mode)
        if not getattr(self.args, "unfreeze_language_model", False):
            self.model.layers.eval()
            self.model.embed_tokens.train(mode)
        if not getattr(self.args, "unfreeze_pc_encoder", False):
            self.model.point_backbone.eval()
        self.engine.pb_train_mode = bool(mode) and bool(getattr(self.args, "unfreeze_pc_encoder", False))
        if self.engine.prepared_bn_stale and not self.engine.pb_train_mode:
            self.engine.fold_stale, self.engine.prepared_bn_stale = True, False     # running stats moved: re-fold BN for eval
        return self
