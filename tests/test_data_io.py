"""CPU: on-disk formats: split JSON, trajectory pickles through the restricted
unpickler, pcrgb .npy, norm_param.json, workspace normalisation round trip."""
import json
import os
import pickle

import numpy as np
import pytest

from egoscaler_amd import data_io as D, traj as T


def _make_dataset(root, data_dir, n=3, style="readme"):
    os.makedirs(data_dir, exist_ok=True)
    images, annots = [], []
    g = np.random.default_rng(0)
    for i in range(n):
        if style == "readme":
            img = {"file_name": f"clip_{i}.5", "take_name": "fair_cooking_07_2", "id": 700 + i}
            take = img["take_name"]
        else:
            img = {"file_name": f"clip_{i}", "dataset_name": "egoexo4d", "video_uid": "uid0", "id": 700 + i}
            take = os.path.join("egoexo4d", "uid0")
        images.append(img)
        annots.append({"image_id": 700 + i, "id": 700 + i, "caption" if style == "readme" else "action_description": f"C Holds The Onion {i}."})
        os.makedirs(os.path.join(root, "pcrgbs", take), exist_ok=True)
        np.save(os.path.join(root, "pcrgbs", take, img["file_name"] + ".npy"), g.normal(size=(600, 6)).astype(np.float32))
        n_steps = 30 + i
        pos = g.uniform([-1, -1, 0.2], [1, 1, 2.0], size=(n_steps, 3))
        rot = g.uniform(-1, 1, size=(n_steps, 3))
        D.save_traj_file(os.path.join(root, "trajs", take, img["file_name"] + (".pkl" if i % 2 == 0 else ".npz")),
                         g.normal(size=(8, 3)), np.concatenate([pos, g.normal(size=(n_steps, 4))], 1), np.concatenate([pos, rot], 1))
    for split in ("train", "val"):
        json.dump({"images": images, "annotations": annots}, open(os.path.join(data_dir, f"{split}.json"), "w"))


@pytest.mark.parametrize("style", ["readme", "dataset_base"])
def test_split_index_and_sample_files(tmp_path, style):
    root, data_dir = str(tmp_path / "EgoScaler"), str(tmp_path / "splits")
    _make_dataset(root, data_dir, style=style)
    f = D.EgoScalerFiles(root, data_dir, "train")
    assert len(f) == 3
    image_id, pc, desc, tr = f.sample(1)
    assert image_id == 701 and pc.shape == (600, 6) and pc.dtype == np.float32
    assert desc == "c holds the onion 1." and tr.shape == (31, 6)
    with pytest.raises(ValueError):
        D.EgoScalerFiles(root, data_dir, "dev")


def test_restricted_unpickler_refuses_code(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))
    p = str(tmp_path / "evil.pkl")
    pickle.dump({"traj_rotvec": Evil()}, open(p, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        D.load_traj_file(p)
    good = str(tmp_path / "t" / "good.pkl")
    D.save_traj_file(good, np.zeros((8, 3)), np.zeros((4, 7)), np.arange(24.0).reshape(4, 6))
    assert np.array_equal(D.load_traj_file(good)["traj_rotvec"], np.arange(24.0).reshape(4, 6))


def test_norm_params_and_workspace_roundtrip(tmp_path):
    D.save_norm_params(str(tmp_path), np.arange(6.0), np.ones(6) * 2)
    m, s = D.load_norm_params(str(tmp_path))
    assert np.array_equal(m, np.arange(6.0)) and np.array_equal(s, np.ones(6) * 2)
    g = np.random.default_rng(1)
    t = np.concatenate([g.uniform([-2, -2, 0], [2, 2, 2.5], size=(2, 7, 3)), g.uniform(-np.pi, np.pi, size=(2, 7, 3))], -1)
    n = D.normalize_workspace(t)
    assert n.min() >= -1 - 1e-12 and n.max() <= 1 + 1e-12
    np.testing.assert_allclose(T.denorm(n), t, atol=1e-12)          # denorm is dataset.py:139-145
