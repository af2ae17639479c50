"""Host-side mirror of the reference's depth wrapper for SURVEY.md §8f row N4
(data/third_party/Depth-Anything-V2/metric_depth/depth.py:13-62, used by data/train/7_get_object_trajectory.py:101-108,244-253).

The network itself (DepthAnythingV2 ViT-L) is a third-party model outside the path: the caller passes any object with the
reference's `infer_image(bgr_uint8_array) -> float32 [h0,w0]` method.  Everything after it — the nearest-neighbour resize to
the frame size, the dense float64 un-projection and colour conversion (get_depth), and the box-masked, depth-thresholded
cloud of `get_points_colors` (pcm_tools.py:68-96) — runs on the MI355X through libegomi.so with no host round trip in
between.  No CPU fallback: without the HIP library every call raises.
"""
import numpy as np
import torch

from . import ops


def _as_array(pil_image):
    return np.asarray(pil_image)            # PIL.Image or ndarray [H,W,3] uint8, as np.array(pil_image) in the reference


class DepthAnything:
    """Same constructor role and method names as the reference class (depth.py:13-62); `model` replaces the checkpoint
    loading of depth.py:16-19 (no weights ship with this build)."""

    def __init__(self, model, device="cuda"):
        self.model, self.device = model, torch.device(device)

    def _predict(self, pil_image):
        image = _as_array(pil_image)
        pred = self.model.infer_image(image[:, :, ::-1])                     # depth.py:25-26,46-47 (RGB -> BGR view)
        pred = torch.as_tensor(np.ascontiguousarray(pred) if isinstance(pred, np.ndarray) else pred, dtype=torch.float32)
        return image, pred.to(self.device)

    @torch.no_grad()
    def get_only_depth(self, pil_image, final_width: int, final_height: int):
        """depth.py:22-32 -> z float32 [final_height, final_width] (numpy)."""
        _, pred = self._predict(pil_image)
        z, _, _ = ops.depth_to_cloud(pred, None, final_width, final_height)
        return z.cpu().numpy()

    @torch.no_grad()
    def get_depth(self, pil_image, final_width: int, final_height: int, focal_len_x: int = 0, focal_len_y: int = 0,
                  principal_point: int = 0):
        """depth.py:35-62 -> (z f32 [H,W], points f64 [H*W,3] | None, colors f64 [H*W,3] | None), numpy like the reference."""
        image, pred = self._predict(pil_image)
        rgb = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        z, pts, col = ops.depth_to_cloud(pred, rgb, final_width, final_height, focal_len_x, focal_len_y, principal_point)
        if pts is None:
            return z.cpu().numpy(), None, None
        return z.cpu().numpy(), pts.cpu().numpy(), col.cpu().numpy()


@torch.no_grad()
def static_scene_cloud(pred, rgb, bbox, principal_p, focal_len_x, focal_len_y, d_thres=None):
    """The per-frame step of 7_get_object_trajectory.py:244-253 kept on the device: resize the predicted depth to the
    frame (N4), then `get_points_colors` with the tracked-object boxes masked out (A1, pcm_tools.py:68-96).
    pred f32 [h0,w0] (device), rgb u8 [H,W,3] (device), bbox = list of {'box': {ymin,ymax,xmin,xmax}} or None.
    -> (points f64 [n,3], colors f32 [n,3]) device tensors in row-major pixel order."""
    H, W = rgb.shape[:2]
    z, _, _ = ops.depth_to_cloud(pred, None, W, H)
    boxes = None if not bbox else [b["box"] for b in bbox]
    pts, col, cnt = ops.unproject_gather(rgb[None, None], z[None, None], principal_p, focal_len_x, focal_len_y, d_thres, boxes)
    n = int(cnt[0])
    return pts[0, :n], col[0, :n]
