"""CPU: host-side glue of rows N1-N3 against vectors recorded from the reference's own code (tests/golden/collate.npz made
by oracle/gen_golden.py:gen_collate from CustomDataset.collate_fn / denorm; demo_trajectory.npz + demo_info.json from the
reference's demo sample by oracle/gen_demo_fixture.py), and the LR schedule against HF's own scheduler (train.py:113-116)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from egoscaler_amd import data_io as D, traj as T
from egoscaler_amd.config import dims_tiny


def test_collate_matches_reference_collate_fn(golden_dir):
    from egoscaler_amd.driver import collate
    g = np.load(os.path.join(golden_dir, "collate.npz"), allow_pickle=False)
    dims = dims_tiny()
    dims.tok.tsep = int(g["tsep"])
    t = lambda k, dt=None: torch.from_numpy(g[k]) if dt is None else torch.from_numpy(g[k]).to(dt)
    out = collate(dims, torch.arange(100, 103), t("pcrgbs"), t("desc"), t("desc_mask"), t("traj_tok"), t("traj_mask"), t("gt"),
                  t("max_abs"), sep_ids=g["sep_ids"].tolist())
    want = {k[4:]: g[k] for k in g.files if k.startswith("out:")}
    assert set(out) == set(want)
    for k, v in want.items():
        got = out[k].numpy()
        assert got.shape == v.shape and got.dtype == v.dtype, (k, got.dtype, v.dtype, got.shape, v.shape)
        assert np.array_equal(got, v), k
    # padding inside the description stays masked in the prompt as well (dataset.py:172-177)
    assert not out["prompt_masks"][1, 6] and not out["attention_masks"][2, 5] and out["attention_masks"][2, 9]


def test_target_norm_denorm_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "collate.npz"), allow_pickle=False)
    x, mabs = g["denorm_in"], g["max_abs"]
    n = T.TargetNorm(do_norm=True)
    got = n.denorm(x, mabs)
    assert got.dtype == g["denorm_norm"].dtype and np.array_equal(got, g["denorm_norm"])
    s = T.TargetNorm(do_standard=True, mean=g["mean"], std=g["std"])
    got = s.denorm(x, mabs)
    assert got.dtype == g["denorm_standard"].dtype and np.array_equal(got, g["denorm_standard"])
    with pytest.raises(AssertionError):
        T.TargetNorm(do_norm=True, do_standard=True)                       # dataset.py:44


def test_target_norm_forward_is_inverse_and_stats_follow_compute_mean_std(tmp_path):
    from oracle import traj as OT
    g = np.random.default_rng(2)
    raws = [np.concatenate([g.uniform([-2, -2, 0], [2, 2, 2.5], size=(n, 3)), g.uniform(-3, 3, size=(n, 3))], 1) for n in (31, 9, 20, 50)]
    # do_norm
    n = T.TargetNorm(do_norm=True)
    v, m = n.normalize(raws[0])
    assert np.abs(v).max() <= 1 + 1e-12 and np.array_equal(m, np.ones(6))
    np.testing.assert_allclose(n.denorm(v[None], m[None])[0], raws[0], atol=1e-12)
    np.testing.assert_allclose(n.denorm(v[None], m[None]), OT.denorm_workspace(v[None]), atol=0)
    # do_standard: statistics == restated compute_mean_std (dataset.py:80-102), side file round trip, inverse
    s = T.TargetNorm(do_standard=True)
    mean, std = s.fit(raws, 20)
    allt = np.array([OT.preprocess_traj(r, 20) for r in raws])
    assert np.array_equal(mean, allt.mean(axis=(0, 1))) and np.array_equal(std, allt.std(axis=(0, 1)) + 1e-8)
    s.save(str(tmp_path))
    assert json.load(open(tmp_path / "norm_param.json")).keys() == {"mean", "std"}
    s2 = T.TargetNorm(do_standard=True).load(str(tmp_path))
    t = OT.preprocess_traj(raws[1], 20)
    v, m = s2.normalize(t)
    assert np.abs(v).max() <= 1 + 1e-12 and np.isclose(np.abs(v).max(0), 1).all()
    np.testing.assert_allclose(s2.denorm(v[None], m[None])[0], t, atol=1e-9)
    with pytest.raises(ValueError):
        T.TargetNorm(do_standard=True).normalize(t)


def test_lr_schedule_equals_hf_linear_schedule_with_warmup():
    from transformers import get_linear_schedule_with_warmup
    from egoscaler_amd.optim import linear_warmup_lr
    for total in (10, 37, 200):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([{"params": [p], "lr": 2e-5}])
        sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=int(total / 5), num_training_steps=total)      # train.py:113-116
        for step in range(total + 3):
            assert linear_warmup_lr(2e-5, step, total) == pytest.approx(sch.get_last_lr()[0], rel=1e-12, abs=1e-20), (total, step)
            opt.step()
            sch.step()
    assert linear_warmup_lr(2e-5, 0, 100) == 0.0 and linear_warmup_lr(2e-5, 100, 100) == 0.0


def test_micro_batch_arithmetic_of_the_deepspeed_config():
    from egoscaler_amd.driver import micro_batch_per_rank
    assert micro_batch_per_rank(64, 1, 8) == 8 and micro_batch_per_rank(8, 1, 1) == 8          # configs[2] / configs[1]
    assert micro_batch_per_rank(8, 4, 1) == 2 and micro_batch_per_rank(8, 3, 1) == 3 and micro_batch_per_rank(2, 4, 8) == 1


def test_reference_demo_sample_reads_through_the_file_layer(golden_dir, tmp_path):
    """The one real EgoScaler record the release holds (assets/demo): trajectory arrays + info.json."""
    z = np.load(os.path.join(golden_dir, "demo_trajectory.npz"), allow_pickle=False)
    info = json.load(open(os.path.join(golden_dir, "demo_info.json")))
    assert set(z.files) == {"init_bbox", "traj", "traj_rotvec"} and z["init_bbox"].shape == (8, 3)
    assert z["traj"].shape == (11, 7) and z["traj_rotvec"].shape == (11, 6) and z["traj_rotvec"].dtype == np.float64
    # same pose track in both parametrisations (7_get_object_trajectory.py:300-328)
    from scipy.spatial.transform import Rotation as R
    assert np.array_equal(z["traj"][:, :3], z["traj_rotvec"][:, :3])
    np.testing.assert_allclose(R.from_quat(z["traj"][:, 3:]).as_rotvec(), z["traj_rotvec"][:, 3:], atol=1e-12)
    # lay it out as the pipeline does (key names exactly as in the demo file: `traj`, not `traj_quat`) and read it back
    root, data_dir = str(tmp_path / "EgoScaler"), str(tmp_path / "splits")
    take, fn = info["take_name"], info["file_name"]
    os.makedirs(os.path.join(root, "trajs", take))
    os.makedirs(os.path.join(root, "pcrgbs", take))
    os.makedirs(data_dir)
    with open(os.path.join(root, "trajs", take, fn + ".pkl"), "wb") as f:
        pickle.dump({k: z[k] for k in ("init_bbox", "traj", "traj_rotvec")}, f)
    np.save(os.path.join(root, "pcrgbs", take, fn + ".npy"), np.random.default_rng(0).normal(size=(600, 6)).astype(np.float32))
    json.dump({"images": [info], "annotations": [info]}, open(os.path.join(data_dir, "test.json"), "w"))
    d = D.load_traj_file(os.path.join(root, "trajs", take, fn + ".pkl"))
    assert np.array_equal(d["traj_quat"], z["traj"]) and np.array_equal(d["traj_rotvec"], z["traj_rotvec"])
    files = D.EgoScalerFiles(root, data_dir, "test")
    image_id, pc, desc, tr = files.sample(0)
    assert image_id == 91 and desc == "c picks up the knife on the counter top with his right hand." and np.array_equal(tr, z["traj_rotvec"])
    # 11 observed steps -> num_steps 20 pads with the last pose (traj_utils.py:21-37); the real track lies inside the workspace
    t20, pm = T.preprocess_traj(tr, 20, return_padding_mask=True)
    assert pm.tolist() == [1] * 11 + [0] * 9 and np.array_equal(t20[11:], np.tile(tr[-1], (9, 1)))
    v, _ = T.TargetNorm(do_norm=True).normalize(t20)
    assert np.abs(v).max() < 1.0
    np.testing.assert_allclose(T.denorm(v[None])[0], t20, atol=1e-12)
    # the npz form is read as well, with the same alias
    assert np.array_equal(D.load_traj_file(os.path.join(golden_dir, "demo_trajectory.npz"))["traj_quat"], z["traj"])


def test_lr_schedule_under_grad_accumulation_documented_deviation():
    """with --grad_accum_steps A > 1 the reference sizes the schedule in loader iterations (A x the optimizer steps,
    train.py:114-116) while DeepSpeed steps it once per accumulation boundary: warm-up A x longer, final rate (1 - 1/A) of the way down,
    never zero.  The build's schedule spans the optimizer steps actually taken (driver.train docstring).  Both statements, against HF."""
    from transformers import get_linear_schedule_with_warmup
    from egoscaler_amd.optim import linear_warmup_lr
    A, opt_steps = 4, 25
    loader_iters = A * opt_steps
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{"params": [p], "lr": 2e-5}])
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=int(loader_iters / 5), num_training_steps=loader_iters)
    ref = []
    for _ in range(opt_steps):
        ref.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    ours = [linear_warmup_lr(2e-5, s, opt_steps) for s in range(opt_steps)]
    assert ref[int(opt_steps / 5)] < 2e-5 * 0.3 and ours[int(opt_steps / 5)] == pytest.approx(2e-5)        # reference still warming up, build at the peak
    assert ref[-1] > 2e-5 * 0.9 and ours[-1] < 2e-5 * 0.1                                               # reference barely decayed, build near zero
    # the reference's totals are reproducible on request: same function, loader-iteration totals
    assert [linear_warmup_lr(2e-5, s, loader_iters) for s in range(opt_steps)] == pytest.approx(ref, rel=1e-12, abs=1e-20)


def test_config_json_written_by_the_reference_classes_parses(golden_dir, tmp_path):
    """VERDICT r2 missing #4: tests/golden/tiny_config.json is the file the reference's own `save_pretrained` wrote for its PointLLMConfig
    (HF PretrainedConfig: `architectures`, `dtype`, `rope_parameters`, `head_dim`, ...; oracle/gen_golden.py::gen_train_steps)."""
    import shutil
    from egoscaler_amd.config import dims_tiny
    from egoscaler_amd.pointllm.model_arch import PointLLMConfig
    shutil.copy(os.path.join(golden_dir, "tiny_config.json"), tmp_path / "config.json")
    with pytest.raises(ValueError, match="tiny.yaml"):
        PointLLMConfig.from_pretrained(str(tmp_path))                       # the name resolves to a YAML, like pointllm.py:38-41
    t = dims_tiny()
    (tmp_path / "tiny.yaml").write_text(
        "model : {\n  NAME: PointTransformer,\n  trans_dim: %d,\n  depth: %d,\n  drop_path_rate: 0.0,\n  cls_dim: 40,\n  num_heads: %d,\n  group_size: %d,\n"
        "  num_group: %d,\n  encoder_dims: %d,\n  point_dims: 3,\n  projection_hidden_layer: 2,\n  projection_hidden_dim: [%d, %d],\n  use_max_pool: false\n}\nnpoints: %d\n"
        % (t.pb.trans_dim, t.pb.depth, t.pb.num_heads, t.pb.group_size, t.pb.num_group, t.pb.encoder_dims, *t.pb.projection_hidden_dim, t.pb.npoints))
    cfg = PointLLMConfig.from_pretrained(str(tmp_path))
    d = cfg.to_dims()
    assert d.lm == t.lm and d.pb == t.pb
    assert (d.tok.pad, d.tok.bos, d.tok.eos) == (0, 1, 2) and cfg.architectures == ["PointLLMLlamaForCausalLM"] and cfg.mm_use_point_start_end is True
    full = PointLLMConfig(point_backbone_config_name="PointTransformer_8192point_2layer")
    full.save_pretrained(str(tmp_path / "full"))
    d7 = PointLLMConfig.from_pretrained(str(tmp_path / "full")).to_dims()
    assert (d7.pb.trans_dim, d7.pb.num_group, d7.pb.group_size, list(d7.pb.projection_hidden_dim)) == (384, 512, 32, [1024, 2048])
    bad = dict(__import__("json").load(open(tmp_path / "config.json")), num_key_value_heads=2)
    (tmp_path / "gqa").mkdir()
    (tmp_path / "gqa" / "config.json").write_text(__import__("json").dumps(bad))
    with pytest.raises(NotImplementedError):
        PointLLMConfig.from_pretrained(str(tmp_path / "gqa"))


def test_ragged_validation_shards_cover_every_sample_once():
    """the reference's val / test loader keeps the short last batch (no drop_last, train.py:79-82)."""
    from egoscaler_amd.dp import ragged_shard_range, shard_range
    for n in range(0, 23):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = ragged_shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                got += list(range(lo, hi))
            assert got == list(range(n))
    assert ragged_shard_range(8, 1, 2) == shard_range(8, 1, 2)
