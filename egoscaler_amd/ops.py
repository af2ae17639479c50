"""Thin torch<->C-ABI marshalling: tensors in, tensors out, raw device pointers underneath.
PyTorch is used for device memory and streams only; every op here runs a kernel of libegomi.so.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import c_p, c_i, c_d, c_f, c_sz, c_i64, call

F32, BF16 = 0, 1


def dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t}")


def P(t):
    if t is None:
        return c_p(None)
    if not t.is_cuda:
        raise _lib.EgomiError("egoscaler_amd ops need CUDA/HIP tensors (no CPU fallback)")
    return c_p(t.data_ptr())


def S():
    return c_p(torch.cuda.current_stream().cuda_stream)


def _c(t, dtype=None):
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------ A1/A2
def unproject_gather(rgb, depth, pp, fx, fy, d_thres=None, boxes=None, n_out=0):
    """rgb u8 [B,T,H,W,3], depth f32 [B,T,H,W] -> (points f64 [B,cap,3], colors f32 [B,cap,3],
    count i32 [B]).  See include/egomi.h (A1)."""
    rgb, depth = _c(rgb, torch.uint8), _c(depth, torch.float32)
    B, T, H, W, _ = rgb.shape
    cap = n_out if n_out > 0 else T * H * W
    dev = rgb.device
    pts = torch.empty(B, cap, 3, dtype=torch.float64, device=dev)
    col = torch.empty(B, cap, 3, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    ws_bytes = _lib.lib().egomi_unproject_workspace_bytes(B, T, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    bx, nb = None, 0
    if boxes is not None and len(boxes):
        bx = torch.tensor([[b["ymin"], b["ymax"], b["xmin"], b["xmax"]] for b in boxes], dtype=torch.int32, device=dev)
        nb = bx.shape[0]
    call("egomi_unproject_gather", P(rgb), P(depth), P(bx), c_i(nb), c_i(B), c_i(T), c_i(H), c_i(W),
         c_d(pp), c_d(fx), c_d(fy), c_f(float("nan") if d_thres is None else d_thres), c_i(n_out),
         P(pts), P(col), P(cnt), P(ws), c_sz(ws_bytes), S())
    return pts, col, cnt


def pc_norm(points, colors):
    points, colors = _c(points, torch.float64), _c(colors, torch.float32)
    B, N, _ = points.shape
    out = torch.empty(B, N, 6, dtype=torch.float32, device=points.device)
    call("egomi_pc_norm", P(points), P(colors), P(out), c_i(B), c_i(N), S())
    return out


# ------------------------------------------------------------------------------------------ A3-A5
def fps(pts, start, num_group):
    pts = _c(pts, torch.float32)
    B, N, C = pts.shape
    start = _c(torch.as_tensor(start, device=pts.device), torch.int32)
    if start.numel() != B or int(start.min()) < 0 or int(start.max()) >= N:
        raise ValueError("fps: start must hold one index in [0,N) per cloud")
    idx = torch.empty(B, num_group, dtype=torch.int32, device=pts.device)
    cen = torch.empty(B, num_group, 3, dtype=torch.float32, device=pts.device)
    call("egomi_fps", P(pts), c_i(B), c_i(N), c_i(C), P(start), c_i(num_group), P(idx), P(cen), S())
    return idx, cen


def knn_group(pts, center, k, out_dtype=torch.float32):
    pts, center = _c(pts, torch.float32), _c(center, torch.float32)
    B, N, C = pts.shape
    G = center.shape[1]
    idx = torch.empty(B, G, k, dtype=torch.int32, device=pts.device)
    nb = torch.empty(B, G, k, C, dtype=out_dtype, device=pts.device)
    call("egomi_knn_group", P(pts), P(center), c_i(B), c_i(N), c_i(C), c_i(G), c_i(k), P(idx), P(nb), c_i(dt(out_dtype)), S())
    return idx, nb
