#!/usr/bin/env python3
"""generate() as run_validation calls it (train.py:223-228: bs 8, prompt up to the first <tsep>, the remaining trajectory positions as new tokens,
sampling defaults) at 7B shapes: seconds per call with a fresh Decoder + graph capture per call (EGOMI_DECODER_CACHE=0, rounds 1-3) against the
cached decoder / replayed graph (round 4).  GPU box only."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b
from egoscaler_amd.pointllm import TrajPointLLMForCausalLM

dims = dims_7b()
if len(sys.argv) > 1:
    dims.lm.num_hidden_layers = int(sys.argv[1])
args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.bfloat16)
with torch.no_grad():
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, p in m.named_parameters():
        if p.dim() >= 2:
            for r0 in range(0, p.shape[0], 8192):
                p.data[r0:r0 + 8192].copy_(torch.empty(p.data[r0:r0 + 8192].shape, dtype=torch.float32, device="cuda").normal_(0, 0.02, generator=g))
        elif "norm" in n and n.endswith("weight"):
            p.data.fill_(1.0)
m.engine.prepared = False
m.eval()
B = 8
toks, masks, Lp = synth.synth_batch(dims, B, text_len=16, num_steps=20, max_traj_token=160)
pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)]).cuda()
T = toks.shape[1] - Lp
print(f"B={B} prompt {Lp} new tokens {T} layers {dims.lm.num_hidden_layers}", flush=True)
for mode in ("0", "1", "0", "1"):
    os.environ["EGOMI_DECODER_CACHE"] = mode
    m.__dict__.pop("_decoders", None)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = m.generate(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts, max_length=T, do_sample=True,
                         fps_start=torch.zeros(B, dtype=torch.int32, device="cuda"), seed=it)
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    print(f"EGOMI_DECODER_CACHE={mode}: seconds per generate() call {[round(t, 3) for t in ts]}  (sequences {tuple(out.sequences.shape)})", flush=True)
