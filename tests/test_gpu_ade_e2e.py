"""GPU: the BASELINE metric's second half, "6DoF ADE vs ref", as ONE chain (train.py:240-260, evaluate.py:128-146):
    generate (prefill + cached greedy decode)  ->  cut at eos, de-tokenise (egomi_traj_detokenize)  ->  rt2 scaling
    ->  pad with the last step  ->  ADE / FDE (egomi_traj_metrics and the as-called host form)
against tests/golden/tiny_trained.npz: a tiny model TRAINED with the reference's own classes (oracle/gen_golden.py:
gen_tiny_trained) until its greedy generations are well-formed but imperfect; the golden holds the reference's greedy ids, the
trajectory its own `str_to_float` parses from them (one step is malformed on purpose of the early stop: the copy-forward rule,
utils.py:88-90, is live) and the ADE / FDE its own metrics.py returns.  Bound: |dADE| <= 1e-3."""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth, traj as T
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu


def test_generate_detokenize_ade_chain_matches_reference(golden_dir):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    g = np.load(os.path.join(golden_dir, "tiny_trained.npz"), allow_pickle=False)
    dims = dims_tiny()
    tok = dims.tok
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32)
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}, strict=True)
    m.eval()
    toks, masks = torch.from_numpy(g["tokens"]), torch.from_numpy(g["masks"])
    Lp, n_new = int(g["prompt_len"]), int(g["n_new"])
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    out = m.generate(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), max_length=n_new,
                     do_sample=False, fps_start=g["fps_start"], eos_token_id=None)     # the golden loop is a fixed-length arg-max loop (gen_golden.py)
    assert np.array_equal(out.sequences.cpu().numpy(), g["gen_sequences"]), "greedy ids differ from the reference's"
    # HF's own loop (the default: eos from the config) pads a finished row and stops when every row has finished: row 0 emits eos one
    # step before row 1, so its last token becomes pad; nothing else changes (generation/utils.py _sample: unfinished_sequences)
    hf = m.generate(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), max_length=n_new + 5,
                    do_sample=False, fps_start=g["fps_start"])
    want = g["gen_sequences"].copy()
    assert want[0, -2] == tok.eos and want[1, -1] == tok.eos
    want[0, -1] = tok.pad
    assert np.array_equal(hf.sequences.cpu().numpy(), want) and len(hf.scores) == n_new
    gen_ids = out.sequences[:, Lp:]
    vals, n = T.detokenize_batch(gen_ids, tok, 8)                       # device: cut at eos, <tsep> segments, copy-forward
    gtv, gn = T.detokenize_batch(toks[:, Lp:].cuda(), tok, 8)
    maxmin = [2.5, 0.1]
    for b in range(2):
        gen = T.rt2_scaler(vals[b, :int(n[b])].cpu().numpy().astype(np.float32), maxmin)          # utils.py:23-34
        gt = T.rt2_scaler(gtv[b, :int(gn[b])].cpu().numpy().astype(np.float32), maxmin)
        assert gen.shape == g[f"gen_traj{b}"].shape and np.array_equal(gen, g[f"gen_traj{b}"]), b      # the parsed trajectory itself is bit-equal
        assert np.array_equal(gt, g[f"gt_traj{b}"])
        ade_called = T.average_displacement_error(gen[None], gt[None])
        assert abs(ade_called - float(g[f"ade_as_called{b}"])) <= 1e-3
        assert abs(T.average_displacement_error(gen, gt) - float(g[f"ade{b}"])) <= 1e-3
        assert abs(T.final_displacement_error(gen, gt) - float(g[f"fde{b}"])) <= 1e-3
        # device metrics kernel (documented [T,D] form), incl. the pad-with-last-step rule on a shortened generation
        gd, gtd = torch.from_numpy(gen)[None].cuda(), torch.from_numpy(gt)[None].cuda()
        ade_d, fde_d = T.metrics_batch(gd, None, gtd)
        assert abs(float(ade_d[0]) - float(g[f"ade{b}"])) <= 1e-3 and abs(float(fde_d[0]) - float(g[f"fde{b}"])) <= 1e-3
        short = torch.tensor([gen.shape[0] - 1], dtype=torch.int32, device="cuda")
        ade_s, _ = T.metrics_batch(gd, short, gtd)
        assert abs(float(ade_s[0]) - float(g[f"ade_short_padded{b}"])) <= 1e-3
    assert float(g["ade0"]) > 0.1                                       # the comparison is not the trivial 0 == 0
