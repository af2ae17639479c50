#!/usr/bin/env python3
"""BASELINE.json config 5: inference-only greedy decode, bs=256, prefill S0~=540 once, then 32
single-token steps captured into ONE hipGraph.  Reports tokens/s and achieved HBM GB/s against the
algorithmic bytes of SURVEY.md §8d ("Roofline - decode": per step 2*P_llm bytes of weights + the K/V
rows read, B*S_ctx*512 KiB).  GPU box only:  python tools/bench_decode.py [--batch 256] [--steps 32]"""
import argparse, json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b
from egoscaler_amd.decode import Decoder
from egoscaler_amd.pointllm import TrajPointLLMForCausalLM



def run(batch=256, steps=32, prefill_chunk=16, layers=None, model=None):
    """-> the JSON record (dict).  Also reachable as `python bench.py --workload decode`, and run by `python bench.py` after the headline
    measurement on the model it already holds (`model`: a frozen-LLM bf16 TrajPointLLMForCausalLM; config.extra.decode)."""
    a = types.SimpleNamespace(batch=batch, steps=steps, prefill_chunk=prefill_chunk, layers=layers)
    dev = torch.device("cuda")
    if model is not None:
        m, dims = model, model.dims
    else:
        dims = dims_7b()
        if a.layers:
            dims.lm.num_hidden_layers = a.layers
        args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=256, model_name=None)
        m = TrajPointLLMForCausalLM(args, dims, None, device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(7)
    with torch.no_grad():
        for n, p in ([] if model is not None else list(m.named_parameters()) + list(m.named_buffers())):
            leaf = n.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                continue
            if leaf == "running_var" or (leaf == "weight" and p.dim() == 1):
                p.fill_(1.0)
            elif leaf == "running_mean":
                p.zero_()
            else:
                fan = p[0].numel() if p.dim() > 1 else p.numel()
                for r0 in range(0, p.shape[0], 8192):
                    blk = p[r0:r0 + 8192]
                    blk.copy_(torch.empty(blk.shape, dtype=torch.float32, device=dev).normal_(0, 0.02 if fan >= 1024 else min(0.35, fan ** -0.5), generator=g))
    m.eval()
    eng = m.engine
    B, T = a.batch, a.steps
    toks, masks, Lp = synth.synth_batch(dims, 1, text_len=16, num_steps=20, max_traj_token=160)
    S0 = Lp
    ids = toks[:, :S0].repeat(B, 1).to(dev)
    pc = synth.synth_cloud(dims, 0)[None].to(dev)
    dec = Decoder(eng, B, S0 + T)
    # prefill in chunks (the prompt pass is not what config 5 times); each chunk fills its slice of the cache
    t0 = time.perf_counter()
    dec.prefill_chunked(ids, None, pc.repeat(B, 1, 1), torch.zeros(B, dtype=torch.int32, device=dev), T, chunk=a.prefill_chunk)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    lg_prefill = dec.lg.clone()
    seq, _ = dec.greedy(T, use_graph=True, keep_scores=False)          # capture + first replay
    torch.cuda.synchronize()
    seq0 = dec.seq.clone()
    reps, ms = 3, 0.0
    for _ in range(reps):
        dec.lg.copy_(lg_prefill)                                       # every replay restarts from the prefill logits
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / reps
    lm = dims.lm
    p_llm = sum(p.numel() for n, p in m.named_parameters() if n.startswith(("model.layers.", "lm_head", "model.norm"))) * 2
    kv_step = [B * (S0 + t + 1) * 2 * lm.hidden_size * 2 * lm.num_hidden_layers for t in range(T - 1)]
    alg_bytes = (T - 1) * p_llm + sum(kv_step)
    out = {"metric": "decode tokens/s (bs=%d, %d steps, hipGraph, greedy)" % (B, T), "value": round(B * T / (ms * 1e-3), 1), "unit": "tokens/s",
           "ms_per_32_steps": round(ms, 2), "ms_per_step": round(ms / max(1, T - 1), 3), "prefill_s": round(t_prefill, 2), "prompt_len": S0,
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg_bytes / (ms * 1e-3) / 8e12, 4), "algorithmic_GB": round(alg_bytes / 1e9, 1)},
           "deterministic_replay": bool(torch.equal(seq0, dec.seq)), "layers": lm.num_hidden_layers}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--prefill-chunk", type=int, default=16, help="prompts are prefilled in chunks of this many samples")
    ap.add_argument("--layers", type=int, default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.steps, a.prefill_chunk, a.layers)))
