"""Stand-in for timm.models.layers (see ../../../README.md). Eval-mode identity only."""
import torch.nn as nn


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob or 0.0)

    def forward(self, x):
        if self.training and self.drop_prob > 0.0:
            raise RuntimeError("DropPath stand-in: stochastic depth is not pinned; use eval()")
        return x
