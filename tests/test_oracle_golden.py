"""CPU: the oracle (oracle/) reproduces every golden vector recorded from the reference itself
(tests/golden/*.npz, made by oracle/gen_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b, dims_tiny
from oracle import llama as OL, pointbert as OPB, pointcloud as OPC, pointllm as OPL, traj as OT


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_unproject_and_pc_norm_bit_exact(golden_dir):
    g = _load(golden_dir, "pointcloud.npz")
    H, W, sid, T = g["meta"]
    rgb, depth = synth.synth_clip(int(sid), int(T), int(H), int(W))
    f, pp = synth.clip_intrinsics(int(H))
    for t in range(int(T)):
        rgbd = np.concatenate([rgb[t], depth[t][..., None]], -1)
        p, c, _ = OPC.unproject_frame(rgbd, W, H, pp, f, f, synth.DEPTH_THRESHOLD)
        assert p.dtype == np.float64 and c.dtype == np.float32
        assert np.array_equal(p, g[f"points{t}"]) and np.array_equal(c, g[f"colors{t}"])
        p, c, _ = OPC.unproject_frame(rgbd, W, H, pp, f, f, synth.DEPTH_THRESHOLD,
                                      [dict(ymin=3, ymax=11, xmin=5, xmax=20)])
        assert np.array_equal(p, g[f"points{t}_box"]) and np.array_equal(c, g[f"colors{t}_box"])
    p, _, _ = OPC.unproject_frame(rgbd, W, H, pp, f, f, None)
    assert np.array_equal(p, g["points_nothres"])
    pc = np.concatenate([g["points0"], g["colors0"].astype(np.float64)], 1)
    assert np.array_equal(OPC.pc_norm(pc), g["pc_norm0"])


def test_traj_known_answers(golden_dir):
    g = _load(golden_dir, "traj.npz")
    v = g["digitize_in"]
    for nb in (256, 16):
        assert np.array_equal(np.array(OT.discretize_action(v, nb)), g[f"digitize_{nb}"])
    # SURVEY.md §8c known answers
    assert OT.discretize_action(np.array([-1, -0.999, 0, 0.5, 1, 1.2, -1.2])) == [0, 0, 127, 191, 255, 255, -1]
    assert np.array_equal(np.array(OT.token_to_action(g["t2a_in"])), g["t2a_out"])
    assert np.array_equal(OT.rt2_scaler(g["rt2_in"].copy(), [2.5, 0.1]), g["rt2_out"])
    s = json.load(open(os.path.join(golden_dir, "traj_strings.json")))["parse_in"]
    assert np.array_equal(OT.rt2_scaler(OT.parse_traj_string(s).copy(), [2.5, 0.1]), g["parse_out"])
    assert OT.parse_traj_string("no tokens") is None
    for T in (50, 20, 7, 3, 2, 1):
        assert np.array_equal(OT.preprocess_traj(g[f"pre_in_{T}"], 20), g[f"pre_out_{T}"])
        assert np.array_equal(OT.smoothing_traj(g[f"pre_in_{T}"]), g[f"smooth_out_{T}"])
    idx = np.linspace(0, 49, 20).astype(int).tolist()
    assert idx == [0, 2, 5, 7, 10, 12, 15, 18, 20, 23, 25, 28, 30, 33, 36, 38, 41, 43, 46, 49]
    assert OT.ade(g["m_gen"], g["m_gt"]) == float(g["ade"])
    assert OT.fde(g["m_gen"], g["m_gt"]) == float(g["fde"])
    assert OT.ade_as_called(g["m_gen20"], g["m_gt"]) == float(g["ade_as_called"])


def test_fps_knn_full_size_indices(golden_dir):
    g = _load(golden_dir, "pointbert_full.npz")
    dims = dims_7b()
    pts = np.stack([synth.synth_cloud(dims, i).numpy() for i in range(2)])
    fidx = OPB.fps_indices(pts[:, :, :3], dims.pb.num_group, g["fps_start"])
    assert np.array_equal(fidx, g["fps_idx"].astype(np.int64)), "FPS indices must be bit-exact"
    center = np.take_along_axis(pts[:, :, :3], fidx[:, :, None].repeat(3, 2), 1)
    assert np.array_equal(center, g["center"])
    k = OPB.knn_indices(pts[:, :, :3], center, dims.pb.group_size)
    ref = g["knn_sets"].astype(np.int64)
    bad = np.argwhere((np.sort(k, -1) != ref).any(-1))
    # The reference's K=3 dot product runs inside a BLAS kernel with unspecified fp32 order, so a
    # group may differ ONLY at a provable near-tie of the k-th / (k+1)-th distance: the expansion
    # -2ab + |a|^2 + |b|^2 rounds at the magnitude of its TERMS, so the bound is 4 ulp of |a|^2+|b|^2.
    d = OPB.square_distance(center, pts[:, :, :3])
    assert len(bad) <= 4
    for b, gi in bad:
        mine, theirs = set(k[b, gi].tolist()), set(ref[b, gi].tolist())
        diff = sorted(mine ^ theirs)
        dd = d[b, gi, diff]
        mag = (center[b, gi] ** 2).sum() + (pts[b, diff, :3] ** 2).sum(-1).max()
        assert len(diff) == 2 and abs(dd[0] - dd[1]) <= 4 * np.spacing(np.float32(mag)), (b, gi, diff, dd)


@pytest.mark.timeout(600)
def test_pointbert_full_features(golden_dir):
    g = _load(golden_dir, "pointbert_full.npz")
    dims = dims_7b()
    sd = {"model.point_backbone." + k: synth.synth_tensor("model.point_backbone." + k, s, 0)
          for k, s in synth.pointbert_param_shapes(dims.pb)}
    pts = synth.synth_cloud(dims, 0)[None]
    with torch.no_grad():
        out = OPB.point_transformer(sd, "model.point_backbone.", pts, dims.pb, g["fps_start"][:1])
    assert out.shape == (1, 513, 384)
    np.testing.assert_allclose(out.numpy(), g["features_b0"], rtol=0, atol=2e-5)


def _tiny(golden_dir):
    g = _load(golden_dir, "tiny_model.npz")
    dims = dims_tiny()
    sd = synth.synth_state_dict(dims, 0)
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    assert np.array_equal(toks.numpy(), g["tokens"]) and np.array_equal(masks.numpy(), g["masks"]) and Lp == int(g["prompt_len"])
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    return g, dims, sd, toks, masks, Lp, pts


def test_tiny_model_forward_loss_grads(golden_dir):
    g, dims, sd, toks, masks, Lp, pts = _tiny(golden_dir)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.startswith("model.point_backbone"))
          for k, v in sd.items()}
    taps = {}
    logits = OPL.forward(sd, dims, toks, masks, pts, g["fps_start"], taps=taps)
    np.testing.assert_allclose(taps["point_features"].detach().numpy(), g["point_features"], rtol=0, atol=1e-5)
    for k in range(dims.lm.num_hidden_layers):
        np.testing.assert_allclose(taps[f"layer{k}"].detach().numpy(), g[f"hidden{k}"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=1e-5)
    loss = OL.traj_loss(logits, toks, Lp, dims.tok.pad)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    got = sorted(k for k, v in sd.items() if v.grad is not None)
    assert got == g["grad_names_all"].tolist()
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), ref, rtol=0, atol=1e-6 + 1e-4 * np.abs(ref).max())


def test_tiny_model_trainable_sets(golden_dir):
    """model_arch.py:33-51: default flags freeze model.layers and the point backbone only."""
    g = _load(golden_dir, "tiny_model.npz")
    fr = g["trainable_frozen_llm"].tolist()
    assert not any(n.startswith(("model.layers.", "model.point_backbone.")) for n in fr)
    assert {"model.embed_tokens.weight", "model.norm.weight", "lm_head.weight", "model.point_proj.0.weight"} <= set(fr)
    un = g["trainable_unfrozen_llm"].tolist()
    assert any(n.startswith("model.layers.") for n in un) and not any(n.startswith("model.point_backbone.") for n in un)


def test_tiny_model_greedy_decode(golden_dir):
    g, dims, sd, toks, masks, Lp, pts = _tiny(golden_dir)
    with torch.no_grad():
        seq, scores = OPL.greedy_generate(sd, dims, toks[:, :Lp], masks[:, :Lp], pts, g["fps_start"], 10)
    assert np.array_equal(seq.numpy(), g["gen_sequences"])
    np.testing.assert_allclose(torch.stack(scores, 1).numpy(), g["gen_scores"], rtol=0, atol=2e-5)


def test_splice_errors(golden_dir):
    g, dims, sd, toks, masks, Lp, pts = _tiny(golden_dir)
    bad = toks.clone()
    bad[0, (bad[0] == dims.tok.point_end).nonzero()[0, 0]] = 5
    with pytest.raises(ValueError) as e:
        OPL.splice_positions(bad, dims.tok, dims.pb.point_token_len)
    assert str(e.value) == str(g["err_missing_end"])
    # text-only sample passes through untouched (pointllm.py:137-142)
    txt = toks.clone()
    txt[1, (txt[1] >= dims.tok.point_patch) & (txt[1] <= dims.tok.point_end)] = 7
    assert OPL.splice_positions(txt, dims.tok, dims.pb.point_token_len)[1] == []


def test_several_segments_per_sample_as_the_reference_treats_them(golden_dir):
    """multi_segment.npz (oracle/gen_golden.py gen_multi_segment, recorded from the reference's own loop, pointllm.py:131-171): two segments
    in sample 0 (only the LAST one spliced, with cloud 0), a text-only sample, one segment in sample 2 (cloud 3 of 4: the running cloud
    index advanced once per segment and once for the text-only sample); one cloud too few -> IndexError."""
    g = _load(golden_dir, "multi_segment.npz")
    dims = dims_tiny()
    sd = synth.synth_state_dict(dims, 0)
    watch = [k[5:] for k in g.files if k.startswith("grad:")]
    sd = {k: v.clone().requires_grad_(k in watch) for k, v in sd.items()}
    toks, masks, Lp = torch.from_numpy(g["tokens"]), torch.from_numpy(g["masks"]), int(g["prompt_len"])
    pts = torch.stack([synth.synth_cloud(dims, 10 + i) for i in range(4)])
    logits = OPL.forward(sd, dims, toks, masks, pts, g["fps_start"])
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=1e-5)
    loss = OL.traj_loss(logits, toks, Lp, dims.tok.pad)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    for n in watch:
        ref = g["grad:" + n]
        np.testing.assert_allclose(sd[n].grad.numpy(), ref, rtol=0, atol=1e-6 + 1e-4 * np.abs(ref).max())
    # the first segment of sample 0 kept its <point_patch> embeddings: the embedding row of that token has a gradient
    assert np.abs(g["grad:model.embed_tokens.weight"][dims.tok.point_patch]).max() > 0
    with pytest.raises(IndexError), torch.no_grad():
        OPL.forward(sd, dims, toks, masks, pts[:3], g["fps_start"][:3])
    assert str(g["err_too_few_clouds"]) == "IndexError"


def test_depth_to_cloud_oracle_vs_golden_and_pillow(golden_dir):
    """N4 oracle: (1) equals the fixture recorded from the reference's get_depth (depth.py:35-62); (2) its index walk equals
    Pillow's NEAREST resize itself on random size pairs (Pillow is the third-party library the reference calls, depth.py:50)."""
    g = np.load(os.path.join(golden_dir, "depth_cloud.npz"))
    for i in range(4):
        rgb = g[f"rgb{i}"]
        H, W = rgb.shape[:2]
        z, p, c = OPC.depth_to_cloud(g[f"pred{i}"], rgb, W, H, float(g[f"f{i}"]), float(g[f"f{i}"]), int(g[f"pp{i}"]))
        assert np.array_equal(z, g[f"z{i}"]) and np.array_equal(p, g[f"points{i}"]) and np.array_equal(c, g[f"colors{i}"])
        assert p.dtype == np.float64 and c.dtype == np.float64 and z.dtype == np.float32
        z0, p0, c0 = OPC.depth_to_cloud(g[f"pred{i}"], rgb, W, H)
        assert p0 is None and c0 is None and np.array_equal(z0, z)
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for _ in range(60):
        h0, w0 = (int(v) for v in rng.integers(2, 400, 2))
        H, W = (int(v) for v in rng.integers(1, 900, 2))
        pred = rng.standard_normal((h0, w0)).astype(np.float32)
        ref = np.array(Image.fromarray(pred).resize((W, H), Image.NEAREST))
        got = pred[OPC.nearest_table(h0, H)][:, OPC.nearest_table(w0, W)]
        assert np.array_equal(ref, got), (h0, w0, H, W)


def test_trained_model_generation_parse_and_ade_chain(golden_dir):
    """tiny_trained.npz (a tiny model trained with the reference's classes): the oracle's cached greedy decode reproduces the
    reference's token ids, its string parser + rt2 scaler the trajectory the reference's `str_to_float` returned, its metrics the
    reference's ADE / FDE — the CPU side of the end-to-end "ADE vs ref" chain (GPU side: tests/test_gpu_ade_e2e.py)."""
    g = _load(golden_dir, "tiny_trained.npz")
    dims = dims_tiny()
    tok = dims.tok
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    toks, masks = torch.from_numpy(g["tokens"]), torch.from_numpy(g["masks"])
    Lp, n_new = int(g["prompt_len"]), int(g["n_new"])
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    with torch.no_grad():
        seq, _ = OPL.greedy_generate(sd, dims, toks[:, :Lp], masks[:, :Lp], pts, g["fps_start"], n_new)
    assert np.array_equal(seq.numpy(), g["gen_sequences"])

    def to_string(ids):
        ids = ids.tolist()
        if tok.eos in ids:
            ids = ids[:ids.index(tok.eos)]
        names = {tok.ts: "<ts>", tok.tsep: "<tsep>", tok.te: "<te>"}
        return " ".join(f"<p{i - tok.p0}>" if tok.p0 <= i < tok.p0 + tok.num_bins else names.get(i, f"<unk{i}>") for i in ids)
    for b in range(2):
        s = to_string(seq[b, Lp:])
        assert s == str(g[f"gen_string{b}"])
        gen = OT.rt2_scaler(OT.parse_traj_string(s, tok.num_bins).copy(), [2.5, 0.1])
        gt = OT.rt2_scaler(OT.parse_traj_string(to_string(toks[b, Lp:]), tok.num_bins).copy(), [2.5, 0.1])
        assert np.array_equal(gen, g[f"gen_traj{b}"]) and np.array_equal(gt, g[f"gt_traj{b}"])
        assert OT.ade(gen, gt) == float(g[f"ade{b}"]) and OT.fde(gen, gt) == float(g[f"fde{b}"])
        assert OT.ade_as_called(gen, gt) == float(g[f"ade_as_called{b}"])
    assert "<tsep> <p" in str(g["gen_string0"]) and float(g["ade0"]) > 0.1


def test_bf16_autocast_golden_is_consistent_with_the_fp32_golden(golden_dir):
    """tiny_model_bf16.npz records the reference under its training numerics (bf16 parameters + autocast) and its distance from the
    fp32 run; the recorded distances must be what the two files imply (they are the bounds of the GPU bf16 tests)."""
    g32, gb = _load(golden_dir, "tiny_model.npz"), _load(golden_dir, "tiny_model_bf16.npz")
    for tag in ("frozen", "unfrozen"):
        rel = abs(float(gb[f"{tag}:loss"]) - float(g32["loss"])) / abs(float(g32["loss"]))
        assert rel == pytest.approx(float(gb[f"{tag}:loss_relerr_vs_fp32"]), rel=1e-9)
        assert 1e-3 < rel < 2e-2
        n = 0
        for k in gb.files:
            if k.startswith(f"{tag}:grad:"):
                name = k[len(tag) + 6:]
                a, b = gb[k].astype(np.float64), g32["grad:" + name].astype(np.float64)
                r = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
                assert r == pytest.approx(float(gb[f"{tag}:relerr_vs_fp32:{name}"]), rel=1e-6) and r < 5e-2
                n += 1
        assert n >= 7
    assert str(gb["frozen:logits_dtype"]) == "torch.bfloat16"
