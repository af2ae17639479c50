#!/usr/bin/env python3
"""Does capturing the whole training step (A1 .. AdamW, ~900 launches) in ONE hipGraph shorten it?  VERDICT r2 #6 estimated 2.5-3.5 ms of
inter-kernel gaps.  Same model / batch as bench.py (configs[1], frozen LLM); eager steps and graph replays alternate in blocks on one box.
GPU box only:  python tools/graph_step_probe.py [--layers N]"""
import argparse, json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from egoscaler_amd import ops, synth
from egoscaler_amd.config import dims_7b
from egoscaler_amd.optim import EgoAdamW
from egoscaler_amd.pointllm import TrajPointLLMForCausalLM

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=None)
ap.add_argument("--mode", default="frozen")
a = ap.parse_args()
dev = torch.device("cuda")
dims = dims_7b()
if a.layers:
    dims.lm.num_hidden_layers = a.layers
B, T, H, W = 8, 8, 224, 224
margs = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=(a.mode == "unfrozen"), num_bins=256, model_name=None)
model = TrajPointLLMForCausalLM(margs, dims, None, device=dev, dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(1234)
with torch.no_grad():
    for n, p in list(model.named_parameters()) + list(model.named_buffers()):
        leaf = n.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            continue
        if leaf == "running_var" or (leaf == "weight" and p.dim() == 1):
            p.fill_(1.0)
        elif leaf == "running_mean":
            p.zero_()
        else:
            fan_in = p[0].numel() if p.dim() > 1 else p.numel()
            std = 0.02 if fan_in >= 1024 else min(0.35, fan_in ** -0.5)
            for r0 in range(0, p.shape[0], 4096):
                blk = p[r0:r0 + 4096]
                blk.copy_(torch.empty(blk.shape, dtype=torch.float32, device=dev).normal_(0, std, generator=g))
model.engine.prepared = False
model.train()
opt = EgoAdamW(model, lr=2e-5)
clips = [synth.synth_clip(i, T, H, W) for i in range(B)]
rgb = torch.from_numpy(np.stack([c[0] for c in clips])).to(dev)
depth = torch.from_numpy(np.stack([c[1] for c in clips])).to(dev)
toks, masks, Lp = synth.synth_batch(dims, B, text_len=16, num_steps=20, max_traj_token=160)
toks, masks = toks.to(dev), masks.to(dev)
fx, pp = synth.clip_intrinsics(H)
fps_start = torch.zeros(B, dtype=torch.int32, device=dev)
N = dims.pb.npoints
loss_buf = torch.zeros((), device=dev)


def step():
    pts, col, cnt = ops.unproject_gather(rgb, depth, pp, fx, fx, synth.DEPTH_THRESHOLD, n_out=N)
    pc = ops.pc_norm(pts, col)
    loss = model.loss_and_backward(toks, masks, pc, Lp, dims.tok.pad, fps_start=fps_start)
    opt.step()
    loss_buf.copy_(loss)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3


for _ in range(4):
    step()
torch.cuda.synchronize()
l_eager = float(loss_buf)
model.engine.defer_splice_check = True
gr = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
t0 = time.perf_counter()
with torch.cuda.stream(side):
    with torch.cuda.graph(gr, stream=side):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t_cap = time.perf_counter() - t0
gr.replay()
torch.cuda.synchronize()
model.engine.check_pending_splice()
res = {"capture_s": round(t_cap, 2), "loss_after_eager_warmup": l_eager, "loss_after_first_replay": float(loss_buf), "rounds": []}
for r in range(3):
    model.engine.defer_splice_check = False
    ev_e, wall_e = timed(step, 10)
    ev_g, wall_g = timed(gr.replay, 10)
    res["rounds"].append({"eager_ms": round(ev_e, 3), "eager_wall_ms": round(wall_e, 3), "graph_ms": round(ev_g, 3), "graph_wall_ms": round(wall_g, 3)})
print(json.dumps(res))
