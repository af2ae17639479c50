"""GPU parity for BASELINE.json configs[0] — "single synthetic 4-frame 224^2 clip + 8-token text, forward only" (the reference's
own CPU-runnable case; SURVEY.md §8d config 1: tiny and 7B-shape): RGB-D frames -> A1 un-projection + subsample -> A2 pc_norm
-> PointBERT -> projector -> splice -> LLaMA -> logits, fp32, against the oracle fed with the same frames."""
import copy
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b, dims_tiny

pytestmark = pytest.mark.gpu


def _run(dims, seed):
    from egoscaler_amd import ops
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    from oracle import pointcloud as OPC, pointllm as OPL
    T, H, W = 4, 224, 224
    rgb, depth = synth.synth_clip(seed, T, H, W)
    f, pp = synth.clip_intrinsics(H)
    N = dims.pb.npoints
    toks, masks, Lp = synth.synth_batch(dims, 1, text_len=8, num_steps=4, max_traj_token=40)
    sd = synth.synth_state_dict(dims, 0)
    # oracle: frames -> cloud -> logits
    pc_o = torch.from_numpy(OPC.clip_to_cloud(rgb, depth, pp, f, synth.DEPTH_THRESHOLD, N))[None]
    with torch.no_grad():
        ref = OPL.forward(sd, dims, toks, masks, pc_o, np.array([0]))
    # device: the same frames through the C-ABI
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, copy.deepcopy(dims), None, device="cuda", dtype=torch.float32)
    m.load_state_dict(sd)
    m.eval()
    pts, col, cnt = ops.unproject_gather(torch.from_numpy(rgb[None]).cuda(), torch.from_numpy(depth[None]).cuda(), pp, f, f, synth.DEPTH_THRESHOLD, n_out=N)
    assert int(cnt[0]) >= N
    pc = ops.pc_norm(pts, col)
    assert np.all(np.abs(pc.cpu().numpy() - pc_o.numpy()) <= np.spacing(np.abs(pc_o.numpy())))          # A1 bit-exact, A2 <= 1 ulp
    with torch.no_grad():
        out = m(input_ids=toks.cuda(), attention_mask=masks.cuda(), point_clouds=pc, fps_start=[0])
    got = out.logits.float().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert got.shape == ref.shape and err < 1e-3, err
    return err


def test_config1_tiny_clip_to_logits():
    _run(dims_tiny(), 11)


@pytest.mark.timeout(600)
def test_config1_7b_shape_clip_to_logits():
    dims = dims_7b()
    dims.lm.num_hidden_layers = 1                      # 7B width, full PointBERT (8192 points, 512 groups), one decoder layer
    _run(dims, 12)
