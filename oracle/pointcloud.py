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


# ---------------------------------------------------------------------------------------------
# N4  depth map -> dense cloud     reference: data/third_party/Depth-Anything-V2/metric_depth/depth.py:35-62
#     (DepthAnything.get_depth after model.infer_image; called from data/train/7_get_object_trajectory.py:101-108)
# ---------------------------------------------------------------------------------------------
def nearest_table(n_src: int, n_dst: int) -> np.ndarray:
    """Source index of every destination index for PIL's Image.resize(..., Image.NEAREST) (depth.py:50).
    Pillow (pinned here: 12.2.0; third-party, its C source is not in /root/reference) walks the row with a running
    double:  xo = a*0.5; for x: idx = (int)xo; xo += a   with a = n_src / n_dst  — the repeated addition, not
    floor((x+0.5)*a), is what its output follows (checked against Pillow itself in tests/test_oracle_golden.py)."""
    a = float(n_src) / float(n_dst)
    out = np.empty(n_dst, dtype=np.int64)
    xo = a * 0.5
    for x in range(n_dst):
        out[x] = int(xo)
        xo += a
    return np.minimum(out, n_src - 1)


def depth_to_cloud(pred: np.ndarray, rgb_u8: np.ndarray, final_width: int, final_height: int,
                   focal_len_x=0, focal_len_y=0, principal_point=0):
    """pred f32 [h0,w0] (network output), rgb u8 [final_height, final_width, 3] ->
    (z f32 [H,W], points f64 [H*W,3] | None, colors f64 [H*W,3] | None), dtypes as numpy yields them at depth.py:51-58."""
    ys, xs = nearest_table(pred.shape[0], final_height), nearest_table(pred.shape[1], final_width)
    z = np.ascontiguousarray(pred[ys][:, xs])                                        # :50-51
    if focal_len_x > 0 and focal_len_y > 0 and principal_point > 0:                  # :53
        x, y = np.meshgrid(np.arange(final_width), np.arange(final_height))
        x = (x - principal_point) / focal_len_x                                      # :55-56 (float64)
        y = (y - principal_point) / focal_len_y
        points = np.stack((np.multiply(x, z), np.multiply(y, z), z), axis=-1).reshape(-1, 3)   # :57
        colors = rgb_u8.reshape(-1, 3) / 255.0                                       # :58 (uint8 / float -> float64)
        return z, points, colors
    return z, None, None
