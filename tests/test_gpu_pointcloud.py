"""GPU parity (through the C-ABI) for SURVEY.md §8a rows A1-A5 against the oracle and the golden
vectors recorded from the reference.  Bit-exact for indices and un-projection; pc_norm within
1 float32 ulp (float64 reduction order differs)."""
import os

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b, dims_tiny

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from egoscaler_amd import ops as O
    return O


def _clip(B, T, H, W, first=0):
    rgb, depth = zip(*[synth.synth_clip(first + i, T, H, W) for i in range(B)])
    return np.stack(rgb), np.stack(depth)


def test_unproject_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "pointcloud.npz"))
    H, W, sid, T = [int(x) for x in g["meta"]]
    rgb, depth = synth.synth_clip(sid, T, H, W)
    f, pp = synth.clip_intrinsics(H)
    for t in range(T):
        r = torch.from_numpy(rgb[t:t + 1][None]).cuda()
        d = torch.from_numpy(depth[t:t + 1][None]).cuda()
        pts, col, cnt = ops.unproject_gather(r, d, pp, f, f, synth.DEPTH_THRESHOLD)
        n = int(cnt[0])
        assert n == g[f"points{t}"].shape[0]
        assert np.array_equal(pts[0, :n].cpu().numpy(), g[f"points{t}"])
        assert np.array_equal(col[0, :n].cpu().numpy(), g[f"colors{t}"])
        pts, col, cnt = ops.unproject_gather(r, d, pp, f, f, synth.DEPTH_THRESHOLD,
                                             boxes=[dict(ymin=3, ymax=11, xmin=5, xmax=20)])
        n = int(cnt[0])
        assert np.array_equal(pts[0, :n].cpu().numpy(), g[f"points{t}_box"])
        assert np.array_equal(col[0, :n].cpu().numpy(), g[f"colors{t}_box"])
    pts, col, cnt = ops.unproject_gather(r, d, pp, f, f, None)
    assert np.array_equal(pts[0, :int(cnt[0])].cpu().numpy(), g["points_nothres"])


@pytest.mark.parametrize("B,T,H,W", [(2, 4, 224, 224), (1, 3, 37, 53), (1, 1, 8, 8)])
def test_unproject_clip_and_subsample_vs_oracle(ops, B, T, H, W):
    from oracle import pointcloud as OPC
    rgb, depth = _clip(B, T, H, W)
    f, pp = synth.clip_intrinsics(H)
    r, d = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
    pts, col, cnt = ops.unproject_gather(r, d, pp, f, f, synth.DEPTH_THRESHOLD)
    n_sub = 8192 if T * H * W > 20000 else 16
    spts, scol, scnt = ops.unproject_gather(r, d, pp, f, f, synth.DEPTH_THRESHOLD, n_out=n_sub)
    for b in range(B):
        po, co = OPC.unproject_clip(rgb[b], depth[b], pp, f, synth.DEPTH_THRESHOLD)
        n = int(cnt[b])
        assert n == po.shape[0] == int(scnt[b])
        assert np.array_equal(pts[b, :n].cpu().numpy(), po) and np.array_equal(col[b, :n].cpu().numpy(), co)
        ps, cs = OPC.strided_subsample(po, co, n_sub)
        assert np.array_equal(spts[b].cpu().numpy(), ps) and np.array_equal(scol[b].cpu().numpy(), cs)
        # pc_norm: float64 sums in a different order -> allow 1 ulp of float32
        ref = OPC.pc_norm(np.concatenate([ps, cs.astype(np.float64)], 1)).astype(np.float32)
        got = ops.pc_norm(spts[b:b + 1], scol[b:b + 1])[0].cpu().numpy()
        assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref))), np.abs(got - ref).max()


def test_unproject_edge_cases(ops):
    # all pixels invalid (zero colour) -> count 0; too few for the subsample -> negative count
    rgb = torch.zeros(1, 1, 16, 16, 3, dtype=torch.uint8, device="cuda")
    depth = torch.ones(1, 1, 16, 16, device="cuda")
    _, _, cnt = ops.unproject_gather(rgb, depth, 7.5, 10.0, 10.0, 5.0)
    assert int(cnt[0]) == 0
    rgb[0, 0, 0, :5] = 9
    _, _, cnt = ops.unproject_gather(rgb, depth, 7.5, 10.0, 10.0, 5.0, n_out=8)
    assert int(cnt[0]) == -5
    depth[0, 0, 0, 2] = float("nan")
    pts, _, cnt = ops.unproject_gather(rgb, depth, 7.5, 10.0, 10.0, 5.0)
    assert int(cnt[0]) == 4          # NaN depth fails z < d_thres, as in numpy


def test_fps_knn_full_size_vs_reference_golden(ops, golden_dir):
    from oracle import pointbert as OPB
    g = np.load(os.path.join(golden_dir, "pointbert_full.npz"))
    dims = dims_7b()
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)]).cuda()
    idx, cen = ops.fps(pts, g["fps_start"], dims.pb.num_group)
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"].astype(np.int32)), "FPS indices must be bit-exact"
    assert np.array_equal(cen.cpu().numpy(), g["center"])
    kidx, nb = ops.knn_group(pts, cen, dims.pb.group_size)
    ko = OPB.knn_indices(pts[:, :, :3].cpu().numpy(), g["center"], dims.pb.group_size)
    assert np.array_equal(kidx.cpu().numpy(), ko.astype(np.int32)), "kNN (distance,index) order must match the oracle bit-exactly"
    # and against the reference's own sets, up to the documented near-tie carve-out (<= 4 groups)
    bad = (np.sort(kidx.cpu().numpy(), -1) != g["knn_sets"].astype(np.int32)).any(-1).sum()
    assert bad <= 4
    nbo, ceno, _, _ = OPB.group(pts.cpu().numpy(), dims.pb.num_group, dims.pb.group_size, g["fps_start"])
    assert np.array_equal(nb.cpu().numpy(), nbo)


@pytest.mark.parametrize("B,N,C,G,K", [(3, 512, 6, 32, 16), (1, 100, 3, 7, 5), (2, 8192, 6, 512, 32), (1, 64, 3, 64, 64)])
def test_fps_knn_shapes_vs_oracle(ops, B, N, C, G, K):
    from oracle import pointbert as OPB
    g = np.random.default_rng(N + C)
    pts = g.normal(size=(B, N, C)).astype(np.float32)
    pts[0, N // 2] = pts[0, N // 3]                 # duplicate point: exact distance tie -> lowest index wins
    start = g.integers(0, N, size=B)
    nbo, ceno, fo, ko = OPB.group(pts, G, K, start)
    p = torch.from_numpy(pts).cuda()
    idx, cen = ops.fps(p, start, G)
    assert np.array_equal(idx.cpu().numpy(), fo.astype(np.int32))
    kidx, nb = ops.knn_group(p, cen, K)
    assert np.array_equal(kidx.cpu().numpy(), ko.astype(np.int32))
    assert np.array_equal(nb.cpu().numpy(), nbo)
    _, nb16 = ops.knn_group(p, cen, K, out_dtype=torch.bfloat16)
    assert torch.equal(nb16.float().cpu(), torch.from_numpy(nbo).bfloat16().float())


def test_bad_arguments_raise(ops):
    from egoscaler_amd._lib import EgomiError
    p = torch.zeros(1, 9000, 3, device="cuda")
    with pytest.raises(EgomiError):
        ops.knn_group(p, torch.zeros(1, 4, 3, device="cuda"), 8)       # N > 8192 unsupported
    with pytest.raises(ValueError):
        ops.fps(torch.zeros(1, 16, 3, device="cuda"), [99], 4)         # start out of range
    with pytest.raises(EgomiError):
        ops.fps(torch.zeros(1, 16, 3), [0], 4)                         # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------------ N4
def test_depth_to_cloud_matches_reference_golden(ops, golden_dir):
    """N4: tests/golden/depth_cloud.npz was recorded by running the reference's DepthAnything.get_depth (depth.py:35-62)
    with a synthetic network output (oracle/gen_golden.py depth_cloud).  Bit-exact: resize indices, f64 points, f64 colours."""
    g = np.load(os.path.join(golden_dir, "depth_cloud.npz"))
    for i in range(4):
        pred, rgb = g[f"pred{i}"], g[f"rgb{i}"]
        H, W = rgb.shape[:2]
        f, pp = float(g[f"f{i}"]), int(g[f"pp{i}"])
        z, pts, col = ops.depth_to_cloud(torch.from_numpy(pred).cuda(), torch.from_numpy(rgb).cuda(), W, H, f, f, pp)
        assert z.dtype == torch.float32 and pts.dtype == torch.float64 and col.dtype == torch.float64
        assert np.array_equal(z.cpu().numpy(), g[f"z{i}"])
        assert np.array_equal(pts.cpu().numpy(), g[f"points{i}"])
        assert np.array_equal(col.cpu().numpy(), g[f"colors{i}"])
        z2, p2, c2 = ops.depth_to_cloud(torch.from_numpy(pred).cuda(), None, W, H)          # get_only_depth / intrinsics 0
        assert p2 is None and c2 is None and torch.equal(z2, z)


@pytest.mark.parametrize("B,h0,w0,H,W", [(2, 518, 518, 1408, 1408), (1, 37, 91, 480, 640), (3, 64, 64, 64, 64), (1, 5, 7, 1, 1)])
def test_depth_to_cloud_vs_oracle(ops, B, h0, w0, H, W):
    """Full Aria frame size (1408^2 from a 518^2 prediction) and odd ratios against the oracle, plus size-independent
    properties: z takes only values of pred, rows/columns repeat monotonically, the cloud's third column is z."""
    from oracle import pointcloud as OPC
    rng = np.random.default_rng(h0 * 1000 + W)
    pred = (rng.random((B, h0, w0), dtype=np.float32) * 5 + 0.1).astype(np.float32)
    rgb = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    f, pp = 610.5, max(1, W // 2)
    z, pts, col = ops.depth_to_cloud(torch.from_numpy(pred).cuda(), torch.from_numpy(rgb).cuda(), W, H, f, f, pp)
    for b in range(B if H * W < 500000 else 1):
        zo, po, co = OPC.depth_to_cloud(pred[b], rgb[b], W, H, f, f, pp)
        assert np.array_equal(z[b].cpu().numpy(), zo)
        assert np.array_equal(pts[b].cpu().numpy(), po)
        assert np.array_equal(col[b].cpu().numpy(), co)
    zz = z.cpu().numpy()
    assert np.array_equal(pts[..., 2].cpu().numpy().reshape(B, H, W), zz.astype(np.float64))
    for b in range(B):
        assert np.isin(zz[b], pred[b]).all()
    ys, xs = OPC.nearest_table(h0, H), OPC.nearest_table(w0, W)
    assert (np.diff(ys) >= 0).all() and (np.diff(xs) >= 0).all() and ys[-1] <= h0 - 1 and xs[-1] <= w0 - 1


def test_depth_to_cloud_bad_arguments(ops):
    from egoscaler_amd._lib import EgomiError
    pred = torch.ones(4, 4, device="cuda")
    with pytest.raises(ValueError):
        ops.depth_to_cloud(pred, torch.zeros(3, 3, 3, dtype=torch.uint8, device="cuda"), 8, 8, 100.0, 100.0, 4)   # rgb size != final size
    with pytest.raises(EgomiError):
        ops.depth_to_cloud(torch.ones(4, 4), None, 8, 8)                                                      # CPU tensor: no fallback


def test_depth_wrapper_mirror_and_static_scene_cloud(ops, golden_dir):
    """The reference-shaped wrapper (depth.py:13-62) returns the golden arrays through its own method names, and the
    device-side composition resize -> get_points_colors(boxes) equals the oracle's two steps."""
    import types
    from egoscaler_amd import depth as D
    from oracle import pointcloud as OPC
    g = np.load(os.path.join(golden_dir, "depth_cloud.npz"))
    pred, rgb = g["pred0"], g["rgb0"]
    H, W = rgb.shape[:2]
    seen = {}

    def infer_image(img):
        seen["bgr"] = np.array_equal(img, rgb[:, :, ::-1])
        return pred
    da = D.DepthAnything(types.SimpleNamespace(infer_image=infer_image))
    z, p, c = da.get_depth(rgb, W, H, focal_len_x=float(g["f0"]), focal_len_y=float(g["f0"]), principal_point=int(g["pp0"]))
    assert seen["bgr"] and isinstance(z, np.ndarray)
    assert np.array_equal(z, g["z0"]) and np.array_equal(p, g["points0"]) and np.array_equal(c, g["colors0"])
    assert np.array_equal(da.get_only_depth(rgb, W, H), g["z0"])
    z2, p2, c2 = da.get_depth(rgb, W, H)
    assert p2 is None and c2 is None
    bbox = [{"box": {"ymin": 4, "ymax": 19, "xmin": 2, "xmax": 30}}, {"box": {"ymin": 30, "ymax": 44, "xmin": 20, "xmax": 36}}]
    f, pp = float(g["f0"]), int(g["pp0"])
    pts, col = D.static_scene_cloud(torch.from_numpy(pred).cuda(), torch.from_numpy(rgb).cuda(), bbox, pp, f, f, d_thres=3.0)
    zo, _, _ = OPC.depth_to_cloud(pred, rgb, W, H)
    rgbd = np.concatenate([rgb, zo[..., None]], -1)
    assert rgbd.dtype == np.float32
    po, co, _ = OPC.unproject_frame(rgbd, W, H, pp, f, f, 3.0, [b["box"] for b in bbox])
    assert np.array_equal(pts.cpu().numpy(), po) and np.array_equal(col.cpu().numpy(), co)
