"""GPU: a batch assembled from on-disk samples (data_io.FileTrajData) feeds the training step."""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import data_io as D, synth
from egoscaler_amd.config import dims_tiny
from tests.test_data_io import _make_dataset

pytestmark = pytest.mark.gpu


def test_file_batch_trains(tmp_path):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny(vocab=320, num_bins=64)
    root, data_dir = str(tmp_path / "EgoScaler"), str(tmp_path / "splits")
    _make_dataset(root, data_dir, n=4)
    files = D.EgoScalerFiles(root, data_dir, "train")
    data = D.FileTrajData(dims, files, encode=lambda s: [3 + (ord(c) % 200) for c in s.split()[0:6] for c in c[:1]], num_steps=5)
    b = data.batch([0, 1, 2, 3], torch.device("cuda"), 48)
    assert b["pcrgbs"].shape == (4, dims.pb.npoints, 6) and b["trajectories"].shape == (4, 5, 6)
    # 'trajectories' are the ground-truth tracks in metres / radians (what train.py:258 compares de-normalised generations with)
    want = np.stack([__import__("egoscaler_amd.traj", fromlist=["x"]).preprocess_traj(files.sample(i)[3], 5) for i in range(4)])
    np.testing.assert_allclose(b["trajectories"].cpu().numpy(), want, rtol=1e-6)
    assert b["max_abs"].shape == (4, 6) and bool(b["attention_masks"][:, 0].all())
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=64, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32)
    m.load_state_dict(synth.synth_state_dict(dims, 0))
    m.train()
    loss = m.loss_and_backward(b["tokens"], b["attention_masks"], b["pcrgbs"], b["prompts"].shape[1], dims.tok.pad, fps_start=[0, 0, 0, 0])
    assert np.isfinite(float(loss))
