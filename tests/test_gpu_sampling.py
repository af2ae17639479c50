"""GPU: egomi_sample_rows (csrc/sample.hip) and generate()'s sampled modes — the reference's DEFAULT generation mode
(models/pointllm/model_arch.py:82-108: do_sample=True, top_k=50, top_p=0.95, temperature, repetition_penalty, output_scores=True).

HF returns the PROCESSED scores, so those are pinned against tests/golden/sampling.npz (HF's own processor objects, built by HF's own
`_get_logits_processor`, applied to the reference model's logits; oracle/gen_golden.py::gen_sampling):
  * kept values bit-exact (the penalty and the temperature are one fp32 multiply / divide each);
  * the -inf pattern identical to HF's except inside ONE group of equal scores on the top-p boundary, where HF's outcome depends on
    torch.sort's unstable tie order (oracle/sampling.py::same_up_to_boundary_ties) — and identical, bit for bit, to the oracle's
    deterministic restatement (lowest index removed first).
The draw cannot be pinned against torch.multinomial (it consumes the global torch RNG); it is checked against the oracle's restatement of
the kernel's own generator (Philox4x32-10, verified against the Random123 known answers in tests/test_sampling_oracle.py) and against
the softmax distribution it has to follow.
"""
import os
import types

import numpy as np
import pytest
import torch

from egoscaler_amd import synth
from egoscaler_amd.config import dims_tiny

pytestmark = pytest.mark.gpu
NEG = float("-inf")


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampling.npz"), allow_pickle=False)
    for name in g["case_names"].tolist():
        T, k, p, rp = g[f"{name}:params"].tolist()
        yield name, torch.from_numpy(g[f"{name}:logits"]), torch.from_numpy(g[f"{name}:input_ids"]), torch.from_numpy(g[f"{name}:scores"]), T, int(k), p, rp


def _run(lg, ids, T, k, p, rp, do_sample=False, seed=0, draw=0, dtype=torch.float32, done=None, eos=None, pad=0):
    from egoscaler_amd.decode import sample_rows
    B, V = lg.shape
    L = ids.shape[1]
    seq = torch.full((B, L + 1), -7, dtype=torch.int64)
    seq[:, :L] = ids
    seq = seq.cuda()
    scores = torch.full((B, V), 123.0, dtype=torch.float32, device="cuda")
    tok = torch.full((B,), -1, dtype=torch.int64, device="cuda")
    rng = torch.tensor([seed, 5], dtype=torch.int64, device="cuda")
    sample_rows(lg.to(dtype).cuda(), scores, seq, L, 0, tok, done, rp, T, k, p, do_sample, rng, draw, eos, pad)
    torch.cuda.synchronize()
    return scores.cpu(), tok.cpu(), seq.cpu()


def test_processed_scores_match_hf_golden(golden_dir):
    from oracle import sampling as OS
    n = 0
    for name, lg, ids, want, T, k, p, rp in _cases(golden_dir):
        got, tok, seq = _run(lg, ids, T, k, p, rp)
        pre = OS.process(lg, ids, rp, T, k, 1.0)
        ok, why = OS.same_up_to_boundary_ties(want, got, pre)
        assert ok, (name, why)
        mine = OS.process(lg, ids, rp, T, k, p)                              # the deterministic restatement: bit-identical, ties included
        assert torch.equal(torch.isinf(got), torch.isinf(mine)), name
        fin = ~torch.isinf(mine)
        assert torch.equal(got[fin], mine[fin]), name
        assert torch.equal(tok, got.argmax(-1)) and torch.equal(seq[:, -1], tok) and torch.equal(seq[:, :-1], ids), name     # greedy: lowest index
        if lg.bfloat16().float().equal(lg):                                  # bf16-representable logits: the bf16 entry gives the same bits
            got16, tok16, _ = _run(lg, ids, T, k, p, rp, dtype=torch.bfloat16)
            assert torch.equal(torch.isinf(got16), torch.isinf(got)) and torch.equal(got16[fin], got[fin]) and torch.equal(tok16, tok), name
        n += 1
    assert n >= 15


def test_draw_follows_the_restated_generator_and_stays_inside_the_kept_set(golden_dir):
    from oracle import sampling as OS
    for name, lg, ids, want, T, k, p, rp in _cases(golden_dir):
        if name not in ("defaults_t0", "all_t4", "ties_k50_p95", "wide_vocab"):
            continue
        for seed, draw in ((1234567890123, 0), (42, 3)):
            got, tok, seq = _run(lg, ids, T, k, p, rp, do_sample=True, seed=seed, draw=draw)
            B, V = got.shape
            noise = torch.from_numpy(OS.gumbel_noise(B, V, seed, 5 + draw))
            pert = got + noise
            best = pert.max(-1).values
            chosen = pert[torch.arange(B), tok]
            assert bool(torch.isfinite(got[torch.arange(B), tok]).all()), name                  # never a removed token
            assert bool(((best - chosen).abs() <= 1e-4).all()), (name, seed, best, chosen)        # the arg-max of scores + noise (logf ulps aside)
            assert torch.equal(seq[:, -1], tok)
        a = _run(lg, ids, T, k, p, rp, do_sample=True, seed=9, draw=1)[1]
        assert torch.equal(a, _run(lg, ids, T, k, p, rp, do_sample=True, seed=9, draw=1)[1])      # same seed and counter: same draw


def test_draw_distribution_is_the_softmax():
    """4096 rows of the same 12 logits, one launch: empirical frequencies within 5 sigma of softmax(logits / T) restricted by top-k."""
    lg1 = torch.tensor([2.0, 1.5, 1.0, 0.5, 0.0, -0.5, -1.0, -1.0, 0.25, 3.0, -4.0, 1.0])
    Bn, T, k = 4096, 0.8, 8
    lg = lg1[None].repeat(Bn, 1)
    ids = torch.zeros(Bn, 1, dtype=torch.int64)
    got, tok, _ = _run(lg, ids, T, k, 1.0, 1.0, do_sample=True, seed=2024, draw=0)
    prob = torch.softmax(got[0], -1)
    assert int(torch.isinf(got[0]).sum()) == 12 - k
    freq = torch.bincount(tok, minlength=12).float() / Bn
    sigma = torch.sqrt(prob * (1 - prob) / Bn)
    assert bool(((freq - prob).abs() <= 5 * sigma + 1e-9).all()), (freq, prob)
    assert float(freq[torch.isinf(got[0])].sum()) == 0.0


def test_eos_and_pad_bookkeeping():
    """HF _sample: `next_tokens = next_tokens * unfinished + pad * (1 - unfinished)`; a row finishes when it emits eos."""
    V = 64
    lg = torch.full((4, V), -1.0)
    lg[0, 9] = lg[1, 2] = lg[2, 9] = lg[3, 2] = 5.0                              # rows 1 and 3 emit eos (= 2)
    done = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device="cuda")         # rows 2 and 3 had finished earlier
    got, tok, seq = _run(lg, torch.zeros(4, 3, dtype=torch.int64), 1.0, 0, 1.0, 1.0, done=done, eos=2, pad=60)
    assert tok.tolist() == [9, 2, 60, 60] and done.cpu().tolist() == [0, 1, 1, 1]
    assert torch.equal(got, lg)                                                  # scores of finished rows are still the model's


def _tiny_model(golden_dir, trained=False):
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    dims = dims_tiny()
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32)
    m.load_state_dict(synth.synth_state_dict(dims, 0))
    toks, masks, Lp = synth.synth_batch(dims, 2, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(2)])
    return m.eval(), dims, toks, masks, Lp, pts


def test_generate_default_sampled_mode_scores_match_hf(golden_dir):
    """generate() with the reference's defaults: .scores[0] are HF's processed scores of the reference model's first-step logits (-inf
    pattern bit-exact — no tie sits on a boundary in these fp32 logits —, kept values <= 1e-3: this model's logits vs the reference's)."""
    g = np.load(os.path.join(golden_dir, "sampling.npz"), allow_pickle=False)
    gm = np.load(os.path.join(golden_dir, "tiny_model.npz"), allow_pickle=False)
    m, dims, toks, masks, Lp, pts = _tiny_model(golden_dir)
    kw = dict(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), fps_start=gm["fps_start"], max_length=5)
    for case, extra in (("defaults_t0", {}), ("all_t0", dict(temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.3))):
        torch.manual_seed(11)
        o = m.generate(**kw, **extra)                                            # do_sample=True, top_k=50, top_p=0.95, T=1.0 by default
        want = torch.from_numpy(g[f"{case}:scores"])
        got = o.scores[0].cpu()
        assert got.dtype == torch.float32 and torch.equal(torch.isinf(got), torch.isinf(want)), case
        fin = ~torch.isinf(want)
        assert float((got[fin] - want[fin]).abs().max()) <= 1e-3 * float(want[fin].abs().max()), case
        assert o.sequences.shape[0] == 2 and Lp < o.sequences.shape[1] <= Lp + 5 and len(o.scores) == o.sequences.shape[1] - Lp
        gen = o.sequences[:, Lp:].cpu()
        for t in range(gen.shape[1]):                                            # every drawn token was a kept one (or pad after eos)
            s = o.scores[t].cpu()
            for b in range(2):
                assert bool(torch.isfinite(s[b, gen[b, t]])) or int(gen[b, t]) == dims.tok.pad
        torch.manual_seed(11)
        o2 = m.generate(**kw, **extra)
        assert torch.equal(o2.sequences, o.sequences)                            # torch.manual_seed makes a sampled run repeatable


def test_sampled_generate_graph_equals_eager(golden_dir):
    gm = np.load(os.path.join(golden_dir, "tiny_model.npz"), allow_pickle=False)
    m, dims, toks, masks, Lp, pts = _tiny_model(golden_dir)
    kw = dict(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), fps_start=gm["fps_start"], max_length=8,
              seed=77, repetition_penalty=1.2, eos_token_id=None)
    a = m.generate(use_graph=True, **kw)
    b = m.generate(use_graph=False, **kw)
    assert torch.equal(a.sequences, b.sequences)
    assert all(torch.equal(x, y) for x, y in zip(a.scores, b.scores))
    c = m.generate(use_graph=True, **{**kw, "seed": 78})
    assert not torch.equal(a.sequences, c.sequences)                             # 8 draws x 2 rows from ~40 kept tokens each: a new seed moves them


def test_generate_rejects_what_hf_rejects(golden_dir):
    m, dims, toks, masks, Lp, pts = _tiny_model(golden_dir)
    kw = dict(input_ids=toks[:, :Lp].cuda(), attention_mask=masks[:, :Lp].cuda(), point_clouds=pts.cuda(), fps_start=[0, 17], max_length=2)
    with pytest.raises(ValueError):
        m.generate(temperature=0.0, **kw)
    with pytest.raises(ValueError):
        m.generate(top_p=1.5, **kw)
    with pytest.raises(ValueError):
        m.generate(repetition_penalty=-1.0, **kw)
