"""GPU: `python -m egoscaler_amd.driver` end to end on EgoScaler FILES from an HF directory (VERDICT r3 missing #2 / weak #6):
`--model_name DIR` -> build_model(args) (config.json + weights + tokenizer of the directory, train.py:67), `--root_dir/--data_dir` ->
EgoScalerFiles + FileTrajData with the tokenizer's encode and the reference's prompt template (dataset.py:16-19), one epoch of training
with a SHORT last batch (train.py:72-77: no drop_last), evaluation, and a `{split}_gen_trajs.json` keyed by IMAGE ID (evaluate.py:150)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny
from tests.test_data_io import _make_dataset
from tests.test_gpu_builder import _make_tokenizer

pytestmark = pytest.mark.gpu


def _hf_dir(path, num_bins):
    """A 'pretrained PointLLM' directory: weights for the base vocabulary + 3 point tokens, config.json, a WordLevel tokenizer."""
    from egoscaler_amd.pointllm import PointLLMConfig, TrajPointLLMForCausalLM
    dims = dims_tiny()
    base_vocab = dims.tok.point_patch
    dims_pt = dims_tiny()
    dims_pt.lm.vocab_size = base_vocab + 3
    cfg = PointLLMConfig.from_dims(dims_pt)
    cfg.vocab_size = base_vocab + 3
    args0 = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=num_bins, model_name=None)
    m0 = TrajPointLLMForCausalLM(args0, dims_pt, None, device="cuda", dtype=torch.float32)
    m0.config = cfg
    m0.load_state_dict(synth.synth_state_dict(dims_pt, 0))
    m0.save_pretrained(path)
    _make_tokenizer(path, base_vocab)
    return base_vocab


def test_cli_trains_and_evaluates_on_files_from_an_hf_directory(tmp_path):
    from egoscaler_amd import driver
    num_bins = 16
    d = str(tmp_path / "PointLLM_tiny")
    base_vocab = _hf_dir(d, num_bins)
    root, data_dir, out = str(tmp_path / "EgoScaler"), str(tmp_path / "splits"), str(tmp_path / "run")
    _make_dataset(root, data_dir, n=5)                                   # image ids 700..704: NOT the dataset indices 0..4
    common = ["--model_name", d, "--root_dir", root, "--data_dir", data_dir, "--bs", "2", "--num_steps", "5", "--max_traj_token", "48",
              "--num_bins", str(num_bins), "--dtype", "fp32", "--do_norm", "--val_greedy", "--out_dir", out]
    logs = []
    orig = driver.train

    def spy(args, model, train_data, val_data=None, device="cuda", log=print, step_log=None):
        # what the command line built: the model of the directory (vocabulary grown by the trajectory tokens), file-backed splits whose
        # description ids come from the directory's tokenizer
        assert model.dims.lm.vocab_size == base_vocab + 3 + 3 + num_bins and model.dims.tok.num_bins == num_bins
        assert type(train_data).__name__ == "FileTrajData" and len(train_data) == 5 and len(val_data) == 5
        b = train_data.batch([0, 1], device, args.max_traj_token)
        assert b["image_ids"].tolist() == [700, 701]
        ids = b["tokens"][0].tolist()
        assert ids[0] == model.dims.tok.bos and model.dims.tok.point_start in ids and model.dims.tok.ts in ids
        return orig(args, model, train_data, val_data, device, log=log, step_log=lambda r: logs.append(r))
    driver.train = spy
    try:
        driver.main(["train", *common, "--epochs", "1"])
    finally:
        driver.train = orig
    # 5 samples at bs 2: three optimizer steps, the last one on the single left-over sample (the reference's loader has no drop_last)
    assert [r["step"] for r in logs] == [0, 1, 2] and all(np.isfinite(r["loss"]) for r in logs)
    assert os.path.exists(os.path.join(out, "latest_model.pt")) and os.path.exists(os.path.join(out, "best_model_ade.pt"))
    driver.main(["eval", *common, "--split", "val", "--checkpoint_dir", out])
    dump = json.load(open(os.path.join(out, "val_gen_trajs.json")))
    assert sorted(int(k) for k in dump) == [700, 701, 702, 703, 704]    # keyed by image id (evaluate.py:150), every sample once
    assert all(np.asarray(v).shape == (5, 6) for v in dump.values())
