"""GPU parity of the MEASURED cached-decode path (BASELINE.json configs[4]) against the CPU oracle: bf16, LLaMA-7B width (d=4096,
ffn=11008, H=32 -> head_dim 128, V=32262), two decoder layers, B=16 prompts of S0=540 tokens (point cloud spliced in) run through
`Decoder.prefill_chunked`, then 4 greedy steps captured in ONE hipGraph.

VERDICT r2 weak #1a: every other 7B-width decode test compares HIP with HIP (fused step == separate kernels, graph == eager,
decode == full forward of the same engine); the only oracle comparison of cached decoding was the tiny fp32 golden, where none of
  * the M <= 512 split-K `gemm_nt_bf16_kernel<128,128,2>` with EGOMI_EPI_SLABS and its consumers `qkv_finish_kernel`
    (sum + RoPE(pos) + cache append) and `slabs_rmsnorm_kernel` (sum + residual + RMSNorm),
  * `attn_decode_kernel` at head_dim 128 against a 540-row cache with a key mask,
  * `prefill_chunked` (per-sample cache appends; chunk = 2 -> M = 1080, the ADVICE r2 range of the GEMM router)
is selected.  A wrong slab stride at d=4096 would have passed everything.  Here the per-step scores and the K/V rows the steps append
are compared with `oracle.pointllm.forward` driven with a KV cache (pointllm.py:255-275, HF modeling_llama.py:243-281), evaluated in
fp32 on the same bf16-rounded weights and TEACHER-FORCED with the tokens the HIP path chose (two logits of random-weight models
are often closer than bf16 noise, so free-running greedy sequences may legitimately fork; what must hold is: same scores within
the bf16 bound at every step, and the same arg-max wherever the oracle's top-2 margin exceeds twice the observed error).

Tolerances as in tests/test_gpu_parity_7b.py (bf16 arithmetic vs fp32 arithmetic; a layout bug is an O(1) error).  Measured on
MI355X: scores Frobenius 0.87e-2 / max 0.9-1.0e-2 at every step, K/V rows 0.3-1.2e-2 / 0.4-2.2e-2, arg-max pinned on 46 of 64 pairs.
"""
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_7b

pytestmark = pytest.mark.gpu

B, T_NEW, CHUNK, LAYERS = 16, 4, 2, 2
FRO_TOL, MAX_TOL = 2.5e-2, 4e-2


def _errs(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30)), float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.timeout(1200)
def test_bf16_7b_width_cached_decode_matches_fp32_oracle():
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    from egoscaler_amd.decode import Decoder
    from oracle import pointllm as OPL
    dims = dims_7b()
    dims.lm.num_hidden_layers = LAYERS
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=16, num_steps=20, max_traj_token=160)
    assert Lp == 540
    ids, pm = toks[:, :Lp].contiguous(), masks[:, :Lp].clone()
    pm[3, 2:4] = False                                       # padding inside two prompts: the key mask reaches attn_decode
    pm[10, 5] = False
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)])
    start = np.arange(B) * 13 % dims.pb.npoints
    sd = synth.synth_state_dict(dims, 0)
    sd = {k: (v.to(torch.bfloat16).float() if v.dtype.is_floating_point else v) for k, v in sd.items()}

    # ---- HIP: chunked prefill, then T_NEW greedy steps replayed from one hipGraph
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.bfloat16)
    m.load_state_dict({k: (v.to(torch.bfloat16) if v.dtype.is_floating_point else v) for k, v in sd.items()}, strict=True)
    m.eval()
    dec = Decoder(m.engine, B, Lp + T_NEW)
    f = dec.fused
    assert f["qkv"] >= 2 and f["o"] >= 2 and f["down"] >= 2, f      # split-K slabs + slab-consuming kernels are what runs at M = 16
    with torch.no_grad():
        dec.prefill_chunked(ids.cuda(), pm.cuda(), pts.cuda(), start, T_NEW, chunk=CHUNK)
        seq, scores = dec.greedy(T_NEW, use_graph=True, keep_scores=True)
    torch.cuda.synchronize()
    assert hasattr(dec, "graph")
    seq = seq.cpu()
    got_scores = [s.float().cpu() for s in scores]
    kc, vc = dec.kc.float().cpu(), dec.vc.float().cpu()                # [L,B,H,Smax,hd]
    assert torch.equal(seq[:, :Lp], ids)

    # ---- oracle: same prompts, fp32, KV cache, fed the tokens the HIP path chose
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    with torch.no_grad():
        cache = [dict() for _ in range(LAYERS)]
        mask = pm.clone()
        lg = OPL.forward(sd, dims, ids, mask, pts, start, cache)[:, -1].float()
        ref_scores = [lg]
        for t in range(1, T_NEW):
            nxt = seq[:, Lp + t - 1:Lp + t]
            mask = torch.cat([mask, torch.ones_like(mask[:, :1])], 1)
            ref_scores.append(OPL.forward(sd, dims, nxt, mask, None, None, cache)[:, -1].float())

    # ---- scores, step by step (step 0 = the chunked prefill's last position; steps >= 1 = cached single-token steps)
    rep = []
    for t in range(T_NEW):
        fro, mx = _errs(got_scores[t], ref_scores[t])
        rep.append((fro, mx))
        assert fro < FRO_TOL and mx < MAX_TOL, (t, fro, mx, rep)
        # the token written by egomi_argmax_rows is the arg-max of the scores the graph recorded (lowest index on ties) ...
        assert torch.equal(seq[:, Lp + t], got_scores[t].argmax(-1)), t
        # ... and the oracle's arg-max wherever the oracle's top-2 margin is beyond what bf16 noise can flip
        err = float((got_scores[t] - ref_scores[t]).abs().max())
        top2 = ref_scores[t].topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2 * err
        assert torch.equal(seq[clear, Lp + t], ref_scores[t].argmax(-1)[clear]), t
    # ---- K/V rows: the prompt rows written by prefill_chunked's per-sample appends and the rows the cached steps appended
    n_rows = Lp + T_NEW - 1                                            # the last chosen token is never fed back
    kv_rep = {}
    for l in range(LAYERS):
        for nm, mine, ref in (("k", kc[l], cache[l]["k"]), ("v", vc[l], cache[l]["v"])):
            assert ref.shape[2] == n_rows
            for part, sl in (("prompt", slice(0, Lp)), ("steps", slice(Lp, n_rows))):
                fro, mx = _errs(mine[:, :, sl], ref[:, :, sl])
                kv_rep[f"{nm}{l}.{part}"] = (fro, mx)
                assert fro < FRO_TOL and mx < MAX_TOL, (l, nm, part, fro, mx)
    assert float(kc[:, :, :, n_rows:].abs().max()) == 0.0               # nothing written past the sequence
    n_clear = int(sum(int(((ref_scores[t].topk(2, -1).values[:, 0] - ref_scores[t].topk(2, -1).values[:, 1]) >
                           2 * float((got_scores[t] - ref_scores[t]).abs().max())).sum()) for t in range(T_NEW)))
    print(f"[decode-parity-7b] fused {f}; scores (fro, max) per step {[(round(a, 4), round(b, 4)) for a, b in rep]}; "
          f"arg-max pinned on {n_clear}/{B * T_NEW} (sample, step) pairs; kv " + "; ".join(f"{k}: {a:.1e}/{b:.1e}" for k, (a, b) in kv_rep.items()))
