"""Stand-in for timm.models.layers (see ../../../README.md).

`DropPath` restates timm 0.4.12's published `drop_path` (the pin of egoscaler/models/pointllm/pyproject.toml; the package is not installed here):
    keep = 1 - p ; r = floor(keep + U[0,1)) per SAMPLE (shape [B, 1, ...]) ; y = x / keep * r        (identity when p == 0 or in eval())
The draw is the one thing that cannot be reproduced, so this stand-in takes it from outside: `MASKS` is a FIFO the fixture generator fills
with the 0/1 tensors `r` (one per call that actually drops, in call order) and every mask used is appended to `USED`; with `MASKS` left at
None a train-mode call with p > 0 still refuses, as before."""
import torch.nn as nn

MASKS = None     # list of [B] tensors of 0./1. to be consumed by the next train-mode calls with p > 0, or None
USED = []


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob or 0.0)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        if MASKS is None:
            raise RuntimeError("DropPath stand-in: stochastic depth is not pinned; fill timm.models.layers.MASKS or use eval()")
        keep = 1.0 - self.drop_prob
        r = MASKS.pop(0).to(x.dtype).reshape((x.shape[0],) + (1,) * (x.ndim - 1))
        USED.append((self.drop_prob, r.flatten().clone()))
        return x.div(keep) * r
