"""Oracle for SURVEY.md §8a rows A3-A8: PointBERT front end (FPS, kNN grouping, mini-PointNet,
12-block ViT encoder).  Functional torch-CPU / numpy code over a state dict that uses the
reference's key names.  Test infrastructure only (see oracle/__init__.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# A3  farthest point sampling      reference: pointbert/misc.py:40-60
# ---------------------------------------------------------------------------------------------
def fps_indices(xyz: np.ndarray, npoint: int, start: np.ndarray) -> np.ndarray:
    """xyz [B,N,3] f32, start [B] -> idx [B,npoint] i64.

    The reference draws `start` from the global torch RNG (misc.py:52); here it is an input.
    fp32 arithmetic in a fixed order: d = (dx*dx + dy*dy) + dz*dz, running min, arg-max with the
    lowest index winning ties (what torch.max returns on CPU; pinned by the golden vectors).
    """
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), dtype=np.int64)
    for b in range(B):
        p = xyz[b]
        dist = np.full((N,), np.float32(1e10), dtype=np.float32)       # misc.py:51
        far = int(start[b])
        for i in range(npoint):
            out[b, i] = far                                            # misc.py:55
            c = p[far]
            dx = p[:, 0] - c[0]
            dy = p[:, 1] - c[1]
            dz = p[:, 2] - c[2]
            d = (dx * dx + dy * dy) + dz * dz                          # misc.py:57
            dist = np.minimum(dist, d)                                 # misc.py:58
            far = int(np.argmax(dist))                                 # misc.py:59
    return out


# ---------------------------------------------------------------------------------------------
# A4  kNN                          reference: pointbert/dvae.py:107-140
# ---------------------------------------------------------------------------------------------
def square_distance(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """[B,S,3],[B,N,3] -> [B,S,N] f32 with the reference's expansion -2ab + |a|^2 + |b|^2
    (dvae.py:137-139), every fp32 operation in a FIXED order (the reference's K=3 dot product runs
    inside a BLAS kernel whose order is unspecified; see knn tie carve-out in the tests):
        dot  = (ax*bx + ay*by) + az*bz
        dist = ((-2*dot) + |a|^2) + |b|^2 ,   |v|^2 = (x*x + y*y) + z*z
    """
    a = np.ascontiguousarray(src, dtype=np.float32)
    b = np.ascontiguousarray(dst, dtype=np.float32)
    dot = (a[:, :, None, 0] * b[:, None, :, 0] + a[:, :, None, 1] * b[:, None, :, 1]) \
        + a[:, :, None, 2] * b[:, None, :, 2]
    na = (a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2]
    nb = (b[..., 0] * b[..., 0] + b[..., 1] * b[..., 1]) + b[..., 2] * b[..., 2]
    d = np.float32(-2.0) * dot
    d = d + na[:, :, None]
    d = d + nb[:, None, :]
    return d.astype(np.float32)


def knn_indices(xyz: np.ndarray, centers: np.ndarray, k: int) -> np.ndarray:
    """k nearest points of each centre -> [B,G,k] i64, ordered by (distance, index) ascending.
    The reference returns the same SET in unspecified order (topk(..., sorted=False), dvae.py:117);
    everything downstream is order-invariant inside a group (max-pool, dvae.py:216,219)."""
    d = square_distance(centers, xyz)
    order = np.argsort(d, axis=-1, kind="stable")       # stable: lower index wins exact ties
    return order[..., :k].astype(np.int64)


# ---------------------------------------------------------------------------------------------
# A5  grouping                     reference: pointbert/dvae.py:150-187
# ---------------------------------------------------------------------------------------------
def group(pts: np.ndarray, num_group: int, group_size: int, start: np.ndarray):
    """pts [B,N,C] (C = 3 or 6) -> neighborhood [B,G,M,C], center [B,G,3], fps idx, knn idx.
    Centres are subtracted from xyz only (dvae.py:182); colour channels are gathered as is."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    B, N, C = pts.shape
    xyz = pts[:, :, :3]
    fidx = fps_indices(xyz, num_group, start)
    center = np.take_along_axis(xyz, fidx[:, :, None].repeat(3, 2), axis=1)      # misc.py:60
    kidx = knn_indices(xyz, center, group_size)
    nb = np.stack([pts[b][kidx[b]] for b in range(B)], 0)                         # [B,G,M,C]
    nb[..., :3] = nb[..., :3] - center[:, :, None, :]
    return nb, center, fidx, kidx


# ---------------------------------------------------------------------------------------------
# A6  mini-PointNet               reference: pointbert/dvae.py:189-221
# ---------------------------------------------------------------------------------------------
def _bn_eval(x, sd, p, eps=1e-5, training=False):
    """training=True: nn.BatchNorm1d in train() mode (batch statistics, running stats updated in place with
    momentum 0.1) — what --unfreeze_pc_encoder gives (model_arch.py:33-36,121-122)."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=training, momentum=0.1, eps=eps)


def pointnet_encoder(sd, prefix: str, neighborhood: torch.Tensor, training=False) -> torch.Tensor:
    """[B,G,M,C] -> [B,G,encoder_dims]; BatchNorm uses running stats (frozen backbone stays in
    eval(), model_arch.py:121-122) unless training=True."""
    B, G, M, C = neighborhood.shape
    x = neighborhood.reshape(B * G, M, C).transpose(2, 1)                        # dvae.py:213-215
    p = prefix + "first_conv."
    x = F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"])
    x = F.relu(_bn_eval(x, sd, p + "1.", training=training))
    x = F.conv1d(x, sd[p + "3.weight"], sd[p + "3.bias"])                        # [BG,256,M]
    g = x.max(dim=2, keepdim=True)[0]                                            # dvae.py:216
    x = torch.cat([g.expand(-1, -1, M), x], dim=1)                               # dvae.py:217
    p = prefix + "second_conv."
    x = F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"])
    x = F.relu(_bn_eval(x, sd, p + "1.", training=training))
    x = F.conv1d(x, sd[p + "3.weight"], sd[p + "3.bias"])
    return x.max(dim=2)[0].reshape(B, G, -1)                                     # dvae.py:219-220


# ---------------------------------------------------------------------------------------------
# A7/A8  transformer encoder      reference: pointbert/point_encoder.py:11-98,169-189
# ---------------------------------------------------------------------------------------------
def vit_attention(sd, p, x, num_heads):
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[p + "qkv.weight"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)                              # point_encoder.py:48
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])


def vit_block(sd, p, x, num_heads, eps=1e-5, drop=None):
    """drop: None, or [2, B] per-sample scales of the two residual branches (DropPath in train mode, point_encoder.py:65,74-75:
    mask / keep_prob with mask = floor(keep_prob + U), timm 0.4.12)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    a = vit_attention(sd, p + "attn.", h, num_heads)
    x = x + (a if drop is None else a * drop[0].to(a.dtype)[:, None, None])      # point_encoder.py:74
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))    # exact erf GELU
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + (h if drop is None else h * drop[1].to(h.dtype)[:, None, None])   # point_encoder.py:75


def point_transformer_from_groups(sd, prefix, neighborhood, center, depth, num_heads, taps=None, training=False, drop=None):
    """Everything after grouping (point_encoder.py:173-186). neighborhood/center: torch f32.
    training=True: BatchNorm in train mode; DropPath through `drop` [depth, 2, B] (per-sample branch scales, see vit_block; None = rate 0)."""
    tok = pointnet_encoder(sd, prefix + "encoder.", neighborhood, training=training)
    if taps is not None:
        taps["pointnet"] = tok
    tok = F.linear(tok, sd[prefix + "reduce_dim.weight"], sd[prefix + "reduce_dim.bias"])
    B = tok.shape[0]
    cls = sd[prefix + "cls_token"].expand(B, -1, -1)
    cls_pos = sd[prefix + "cls_pos"].expand(B, -1, -1)
    pos = F.linear(F.gelu(F.linear(center, sd[prefix + "pos_embed.0.weight"], sd[prefix + "pos_embed.0.bias"])),
                   sd[prefix + "pos_embed.2.weight"], sd[prefix + "pos_embed.2.bias"])
    x = torch.cat((cls, tok), dim=1)
    pos = torch.cat((cls_pos, pos), dim=1)
    if taps is not None:
        taps["x0"], taps["pos"] = x, pos
    for i in range(depth):
        x = vit_block(sd, f"{prefix}blocks.blocks.{i}.", x + pos, num_heads, drop=None if drop is None else drop[i])     # pos re-added: :95-98
        if taps is not None:
            taps[f"block{i}"] = x
    C = x.shape[-1]
    return F.layer_norm(x, (C,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)


def point_transformer(sd, prefix, pts: torch.Tensor, pb, start, taps=None, training=False, drop=None):
    """pts [B,N,C] f32 -> [B,G+1,trans_dim] (use_max_pool=false: all tokens, point_encoder.py:186-187)."""
    nb, center, fidx, kidx = group(pts.detach().numpy(), pb.num_group, pb.group_size, np.asarray(start))
    if taps is not None:
        taps["fps_idx"], taps["knn_idx"] = fidx, kidx
        taps["neighborhood"], taps["center"] = nb, center
    return point_transformer_from_groups(sd, prefix, torch.from_numpy(nb), torch.from_numpy(center),
                                         pb.depth, pb.num_heads, taps, training=training, drop=drop)
