"""GPU: the train / evaluate drivers end to end on the seeded synthetic data source (tiny model):
loss goes down, validation by generation returns finite ADE/FDE, the checkpoint has the reference's
dict layout (train.py:287-308) and --resume continues from it, evaluate() dumps {split}_gen_trajs.json."""
import json
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu


def _setup(tmp, dtype=torch.float32, epochs=3):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny(vocab=320, num_bins=64)
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=True, num_bins=64, model_name=None,
                                 max_traj_token=48, num_steps=5, epochs=epochs, bs=4, lr_llm=3e-3, resume=False, out_dir=str(tmp),
                                 checkpoint_dir=str(tmp), val_batches=1)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=dtype)
    m.load_state_dict(synth.synth_state_dict(dims, 0))
    return dims, args, m


def test_build_batch_contract():
    from egoscaler_amd.driver import SyntheticTrajData
    dims = dims_tiny(vocab=320, num_bins=64)
    data = SyntheticTrajData(dims, 8, frames=2, size=32, text_len=8, num_steps=5)
    b = data.batch([0, 1, 2], torch.device("cuda"), 48)
    assert set(b) == {"image_ids", "pcrgbs", "prompts", "prompt_masks", "tokens", "attention_masks", "trajectories", "trajectory_masks", "max_abs"}
    tok, P = dims.tok, dims.pb.point_token_len
    t0 = b["tokens"][0].tolist()
    s = t0.index(tok.point_start)
    assert t0[0] == tok.bos and t0[s + P + 1] == tok.point_end and t0.count(tok.point_patch) == P
    Lp = b["prompts"].shape[1]
    assert t0[Lp - 1] == tok.tsep and t0[Lp - 8] == tok.ts               # prompt = ... <ts> first step <tsep>  (dataset.py:180-182)
    assert b["pcrgbs"].shape == (3, dims.pb.npoints, 6) and b["tokens"].shape == b["attention_masks"].shape
    assert bool(b["attention_masks"][0, :Lp].all()) and not bool(b["attention_masks"][0, -1])


def test_train_validate_checkpoint_resume_evaluate(tmp_path):
    from egoscaler_amd.driver import SyntheticTrajData, train, evaluate
    dims, args, model = _setup(tmp_path)
    tr = SyntheticTrajData(dims, 16, frames=2, size=32, text_len=8, num_steps=5)
    va = SyntheticTrajData(dims, 4, frames=2, size=32, text_len=8, num_steps=5, seed=977)
    hist = train(args, model, tr, va, torch.device("cuda"), log=lambda s: None)
    assert len(hist) == 3 and hist[-1]["train_loss"] < hist[0]["train_loss"] - 0.2, hist
    assert all(np.isfinite(h["ADE"]) and np.isfinite(h["FDE"]) for h in hist)
    ck = torch.load(os.path.join(tmp_path, "latest_model.pt"), map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "global_step"} <= set(ck)
    assert ck["epoch"] == 2 and ck["global_step"] == 3 * (16 // 4)
    assert list(ck["model_state_dict"].keys()) == [k for k, _ in synth.param_shapes(dims)]
    assert os.path.exists(os.path.join(tmp_path, "best_model_ade.pt"))
    # resume: a fresh model picks up epoch 3 and keeps improving on the same data
    dims2, args2, model2 = _setup(tmp_path, epochs=4)
    args2.resume = True
    hist2 = train(args2, model2, tr, va, torch.device("cuda"), log=lambda s: None)
    # (a model restarted from the seeded weights would be back at hist[0]'s loss)
    assert [h["epoch"] for h in hist2] == [3] and hist2[0]["train_loss"] < hist[0]["train_loss"] - 0.1, (hist, hist2)
    m = evaluate(args2, model2, va, "test", torch.device("cuda"))
    assert np.isfinite(m["ADE"]) and m["n"] == 4
    dump = json.load(open(os.path.join(tmp_path, "test_gen_trajs.json")))
    assert len(dump) == 4 and len(next(iter(dump.values()))["gen_traj"][0]) == 6
