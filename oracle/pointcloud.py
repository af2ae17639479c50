"""Oracle for SURVEY.md §8a rows A1 (RGB-D un-projection + compaction) and A2 (pc_norm). numpy.

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np


def unproject_frame(rgbd: np.ndarray, width: int, height: int, pp: float, fx: float, fy: float,
                    d_thres=None, boxes=None):
    """Follows egoscaler/data/tools/pcm_tools.py:68-96 (get_points_colors).

    rgbd [H,W,4]: channels 0..2 colour (0..255 valued), channel 3 depth.  Returns (points [n,3],
    colors [n,3], valid [H*W] bool) in row-major pixel order.  dtype behaviour is numpy's: the
    integer meshgrid divided by a python float is float64, so points are float64; colors keep the
    dtype of rgbd (float32 in / float32 out).
    """
    image = rgbd[:, :, :3]
    z = rgbd[:, :, 3]
    u, v = np.meshgrid(np.arange(width), np.arange(height))
    xn = (u - pp) / fx                                   # :74-75
    yn = (v - pp) / fy
    pts = np.stack((xn * z, yn * z, z), axis=-1).reshape(-1, 3)   # :77
    cols = image.reshape(-1, 3) / 255.0                  # :78
    valid = np.all(image != 0, axis=2)                   # :79
    if boxes is not None:                                # :80-85 (ymin:ymax, xmin:xmax zeroed)
        keep = np.ones((height, width), dtype=bool)
        for b in boxes:
            keep[b["ymin"]:b["ymax"], b["xmin"]:b["xmax"]] = False
        valid = valid & keep
    if d_thres is not None:                              # :87-89
        valid = valid & (z < d_thres)
    valid = valid.ravel()
    return pts[valid], cols[valid], valid


def unproject_clip(rgb_u8: np.ndarray, depth: np.ndarray, pp: float, f: float, d_thres: float,
                   boxes=None):
    """All T frames, frame-major then row-major. rgb u8 [T,H,W,3], depth f32 [T,H,W].
    The rgbd concatenation mirrors vis/interactive.py:22-32 (uint8 + float32 -> float32)."""
    T, H, W, _ = rgb_u8.shape
    P, C = [], []
    for t in range(T):
        rgbd = np.concatenate([rgb_u8[t], depth[t][..., None]], axis=-1)
        assert rgbd.dtype == np.float32
        p, c, _ = unproject_frame(rgbd, W, H, pp, f, f, d_thres, boxes)
        P.append(p)
        C.append(c)
    return np.concatenate(P, 0), np.concatenate(C, 0)


def strided_subsample(points: np.ndarray, colors: np.ndarray, n: int):
    """First-N-valid strided subsample: stride = floor(n_valid / N), deterministic."""
    nv = points.shape[0]
    if nv < n:
        raise ValueError(f"only {nv} valid points, need {n}")
    stride = nv // n
    idx = np.arange(n) * stride
    return points[idx], colors[idx]


def pc_norm(pc: np.ndarray) -> np.ndarray:
    """Follows pointllm/pointllm/data/utils.py:146-157: centre xyz on its centroid, scale by the
    largest radius; other channels untouched."""
    xyz = pc[:, :3]
    other = pc[:, 3:]
    centroid = np.mean(xyz, axis=0)
    xyz = xyz - centroid
    m = np.max(np.sqrt(np.sum(xyz ** 2, axis=1)))
    xyz = xyz / m
    return np.concatenate((xyz, other), axis=1)


def clip_to_cloud(rgb_u8, depth, pp, f, d_thres, n_points):
    """clip -> [N,6] float32 cloud, the build-defined glue of SURVEY.md §8d."""
    p, c = unproject_clip(rgb_u8, depth, pp, f, d_thres)
    p, c = strided_subsample(p, c, n_points)
    pc = np.concatenate([p, c.astype(np.float64)], axis=1)
    return pc_norm(pc).astype(np.float32)
