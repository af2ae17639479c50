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
                                 checkpoint_dir=str(tmp), val_batches=1, val_sample=False, grad_accum_steps=1)
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
    assert len(dump) == 4 and len(next(iter(dump.values()))[0]) == 6          # {image_id: [[x,y,z,rx,ry,rz], ...]} (evaluate.py:150)
    assert {"ADE", "FDE", "GD", "ADE_as_called"} <= set(m)
    # the reference validates by SAMPLING (model_arch.py:82-88 defaults): that path runs too
    args2.val_sample = True
    ms = evaluate(args2, model2, va, "test", torch.device("cuda"))
    assert ms["n"] == 4 and np.isfinite(ms["FDE"])


def test_backward_overwrites_unless_accumulating():
    """from step 2 on the fp32 driver loop summed gradients over steps.  loss_and_backward overwrites unless
    `accumulate_grads` is set, in both dtypes; two micro-batches accumulated == one batch of twice the size."""
    from egoscaler_amd.driver import SyntheticTrajData
    for dtype in (torch.float32, torch.bfloat16):
        dims, args, m = _setup("/tmp", dtype=dtype)
        m.train()
        data = SyntheticTrajData(dims, 8, frames=2, size=32, text_len=8, num_steps=5)
        dev = torch.device("cuda")
        b = data.batch([0, 1, 2, 3], dev, 48)
        z = torch.zeros(4, dtype=torch.int32, device=dev)
        call = lambda bb, zz: m.loss_and_backward(bb["tokens"], bb["attention_masks"], bb["pcrgbs"], bb["prompts"].shape[1], dims.tok.pad, fps_start=zz)
        call(b, z)
        g1 = {n: g.clone() for n, g in m.engine.main_grad.items()}
        call(b, z)                                                     # second step, same batch: must equal a single fresh backward
        for n, g in m.engine.main_grad.items():                        # (the embedding gradient is a scatter of atomic adds: equal up to fp32 summation order)
            assert torch.allclose(g, g1[n], rtol=1e-5, atol=1e-6 * float(g1[n].abs().max()) + 1e-12), (dtype, n)
        m.accumulate_grads = True
        call(b, z)
        for n, g in m.engine.main_grad.items():
            assert torch.allclose(g, 2 * g1[n], rtol=1e-5, atol=1e-7), (dtype, n)
        m.accumulate_grads = False
        # micro-batches [0,1] + [2,3] accumulate to 2 x the mean gradient of the batch of four (equal token counts)
        h0, h1 = data.batch([0, 1], dev, 48), data.batch([2, 3], dev, 48)
        call(h0, z[:2])
        m.accumulate_grads = True
        call(h1, z[:2])
        m.accumulate_grads = False
        tol = 1e-4 if dtype == torch.float32 else 4e-2
        for n in ("lm_head.weight", "model.point_proj.4.weight", "model.layers.0.self_attn.q_proj.weight"):
            a, ref = m.engine.main_grad[n] * 0.5, g1[n]
            assert float((a - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-8, (dtype, n)


def test_grad_accum_steps_equals_one_big_batch(tmp_path):
    """--grad_accum_steps 2 (micro-batches of 2) takes the same optimizer step as --grad_accum_steps 1 (batch of 4): train.py:85,94-96."""
    from egoscaler_amd.driver import SyntheticTrajData, train
    after = {}
    for accum in (1, 2):
        dims, args, model = _setup(tmp_path / f"a{accum}", epochs=1)
        args.grad_accum_steps, args.bs, args.lr_llm = accum, 4, 1e-3
        tr = SyntheticTrajData(dims, 4, frames=2, size=32, text_len=8, num_steps=5)
        hist = train(args, model, tr, None, torch.device("cuda"), log=lambda s: None)
        assert hist[0]["global_step"] == 1
        after[accum] = ({n: p.detach().clone() for n, p in model.named_parameters()}, hist[0]["train_loss"])
    assert abs(after[1][1] - after[2][1]) < 1e-4 * abs(after[1][1])
    moved = 0
    ref = synth.synth_state_dict(dims_tiny(vocab=320, num_bins=64), 0)
    for n, p in after[1][0].items():
        d = float((p - after[2][0][n]).abs().max())
        step = float((p.cpu() - ref[n]).abs().max())
        moved += step > 0
        assert d <= 0.02 * step + 1e-7, (n, d, step)                  # same AdamW step (sign-like first step: tiny gradient differences move little)
    assert moved > 20


def test_ragged_descriptions_are_masked_and_do_standard_round_trips(tmp_path):
    """N1/N2: --max_desc_token padding carries mask False into attention_masks / prompt_masks (dataset.py:161-177), the
    masked filler ids do not influence the loss; --do_standard statistics are fitted on the train split, written to
    norm_param.json (dataset.py:58-124) and undo the normalisation of the tokenised targets up to the bin width."""
    from egoscaler_amd import traj as T
    from egoscaler_amd.driver import SyntheticTrajData, train, evaluate
    dims, args, model = _setup(tmp_path, epochs=1)
    norm = T.TargetNorm(do_standard=True)
    tr = SyntheticTrajData(dims, 8, frames=2, size=32, text_len=8, num_steps=5, norm=norm, max_desc_token=12, ragged_text=True)
    va = SyntheticTrajData(dims, 4, frames=2, size=32, text_len=8, num_steps=5, seed=977, norm=norm, max_desc_token=12, ragged_text=True)
    hist = train(args, model, tr, va, torch.device("cuda"), log=lambda s: None)
    assert np.isfinite(hist[0]["ADE"]) and os.path.exists(os.path.join(tmp_path, "norm_param.json"))
    dev = torch.device("cuda")
    b = tr.batch([0, 1, 2, 3], dev, 48)
    am = b["attention_masks"]
    P = dims.pb.point_token_len
    assert am.dtype == torch.int64 and int((am[:, :P + 15] == 0).sum()) > 0 and bool(am[:, 0].all())       # some description padding is masked
    assert torch.equal(b["prompt_masks"], am[:, :b["prompts"].shape[1]])
    # masked positions: any token id there gives the same loss
    model.train()
    z = torch.zeros(4, dtype=torch.int32, device=dev)
    l0 = float(model.loss_and_backward(b["tokens"], am, b["pcrgbs"], b["prompts"].shape[1], dims.tok.pad, fps_start=z, backward=False))
    t2 = b["tokens"].clone()
    head = torch.zeros_like(am, dtype=torch.bool)
    head[:, :P + 15] = True
    t2[(am == 0) & head] = 7
    l1 = float(model.loss_and_backward(t2, am, b["pcrgbs"], b["prompts"].shape[1], dims.tok.pad, fps_start=z, backward=False))
    assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
    # targets: tokens -> values -> denorm == ground truth within one bin of the standardised scale
    Lp = b["prompts"].shape[1]
    vals, n = T.detokenize_batch(b["tokens"][:, Lp - 7:], dims.tok, 9)
    assert n.tolist() == [6] * 4                                          # 5 steps + the trailing "<te>" segment, which copies the last step (utils.py:88-90)
    assert torch.equal(vals[:, 5], vals[:, 4])
    rec = norm.denorm(vals[:, :5].cpu().numpy(), b["max_abs"].cpu().numpy())
    gt = b["trajectories"].cpu().numpy()
    binw = 2.0 / (dims.tok.num_bins - 1)
    bound = binw * b["max_abs"].cpu().numpy()[:, None, :] * norm.std[None, None, :]
    assert (np.abs(rec - gt) <= bound + 1e-5).all()
    # evaluate(): a fresh TargetNorm picks the statistics up from checkpoint_dir/norm_param.json (evaluate.py:91)
    va2 = SyntheticTrajData(dims, 4, frames=2, size=32, text_len=8, num_steps=5, seed=977, norm=T.TargetNorm(do_standard=True), max_desc_token=12, ragged_text=True)
    m = evaluate(args, model, va2, "val", dev)
    assert m["n"] == 4 and np.allclose(va2.norm.mean, norm.mean)


@pytest.mark.parametrize("unfreeze", [False, True], ids=["frozen_llm", "unfrozen_llm"])
def test_driver_train_reproduces_the_reference_training_trajectory(golden_dir, tmp_path, unfreeze):
    """VERDICT r2 missing #5: the COMPOSITION of the reference's loop over several optimizer steps (train.py:107-117 torch AdamW defaults,
    weight_decay 0.01, HF linear schedule; :157-184 zero_grad / forward / span CE / backward / step), recorded from the reference's own
    classes in oracle/gen_golden.py::gen_train_steps: per-step loss, per-step learning rate and the weights after 6 steps (2 epochs x 3
    batches) must come out of `egoscaler_amd.driver.train` itself — fp32, the same batches — within 1e-3."""
    from egoscaler_amd.driver import train
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    g = np.load(os.path.join(golden_dir, "train_steps.npz"), allow_pickle=False)
    tag = "unfrozen" if unfreeze else "frozen"
    dims = dims_tiny()
    K, Lp = int(g["K"]), int(g["prompt_len"])
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=unfreeze, num_bins=dims.tok.num_bins, model_name=None,
                                 max_traj_token=40, num_steps=4, epochs=2, bs=2, lr_llm=float(g["lr"]), resume=False, out_dir=str(tmp_path),
                                 checkpoint_dir=str(tmp_path), grad_accum_steps=1)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.float32)
    sd0 = synth.synth_state_dict(dims, 0)
    m.load_state_dict(sd0)

    class RecordedBatches:
        """The three recorded batches in the recorded order, whatever permutation the driver draws (the golden's loader order is its own)."""
        def __init__(self):
            self.i = 0

        def __len__(self):
            return 3 * args.bs

        def batch(self, idx, device, max_traj_token):
            j = self.i % 3
            self.i += 1
            toks = torch.from_numpy(g[f"tokens{j}"]).to(device)
            return {"tokens": toks, "attention_masks": torch.from_numpy(g[f"masks{j}"]).to(device), "pcrgbs": torch.from_numpy(g[f"points{j}"]).to(device),
                    "prompts": toks[:, :Lp]}
    steps = []
    hist = train(args, m, RecordedBatches(), None, torch.device("cuda"), log=lambda s: None, step_log=steps.append)
    assert len(steps) == K and [s["step"] for s in steps] == list(range(K))
    want_l, want_lr = g[f"{tag}:losses"], g[f"{tag}:lrs"]
    got_l = np.array([s["loss"] for s in steps])
    assert np.allclose([s["learning_rate"] for s in steps], want_lr, rtol=1e-12, atol=1e-20), (steps, want_lr)
    assert float(np.abs(got_l - want_l).max() / np.abs(want_l).max()) < 1e-3, (got_l, want_l)
    assert want_l[0] - want_l[-1] > 1.0                                                   # the trajectory is not a constant
    assert abs(hist[0]["train_loss"] - float(want_l[:3].mean())) < 1e-3 * float(want_l[:3].mean())        # per-epoch mean, train.py:186,272
    sd = m.state_dict()
    rep = {}
    for k in g.files:
        if k.startswith(f"{tag}:w:"):
            n = k[len(tag) + 3:]
            ref, got, init = g[k], sd[n].float().cpu().numpy(), sd0[n].numpy()
            moved = float(np.abs(ref - init).max())
            err = float(np.abs(got - ref).max())
            rep[n] = (err / (float(np.abs(ref).max()) + 1e-30), err / (moved + 1e-30))
            assert rep[n][0] < 1e-3, (n, rep[n])                                          # the north-star tolerance on the tensor ...
            assert rep[n][1] < 2e-2, (n, rep[n])                                          # ... and the UPDATE itself reproduced to 2 % of its size
    assert len(rep) >= (8 if unfreeze else 5)
    if not unfreeze:
        assert torch.equal(sd["model.layers.0.self_attn.q_proj.weight"].cpu(), sd0["model.layers.0.self_attn.q_proj.weight"])
    print(f"[train-steps {tag}] losses {np.round(got_l, 5).tolist()} vs {np.round(want_l, 5).tolist()}; (rel err, err / movement): "
          + "; ".join(f"{n.replace('model.', '')} {a:.1e}/{b:.1e}" for n, (a, b) in rep.items()))


def test_validation_keeps_the_short_last_batch(tmp_path):
    """len(data) % bs != 0 — every sample is generated for and every image_id lands in the dump (the reference's
    val / test DataLoader has no drop_last); len(data) < bs still evaluates."""
    from egoscaler_amd.driver import SyntheticTrajData, evaluate
    dims, args, model = _setup(tmp_path)
    args.val_batches = None
    seen = []
    gen = model.generate
    model.generate = lambda **kw: (seen.append(int(kw["input_ids"].shape[0])), gen(**kw))[1]
    for n in (6, 3):                                                      # bs = 4: one full + one short batch; a single short batch
        seen.clear()
        va = SyntheticTrajData(dims, n, frames=2, size=32, text_len=8, num_steps=5, seed=977)
        m = evaluate(args, model, va, "test", torch.device("cuda"))
        assert seen == ([4, 2] if n == 6 else [3]), seen                   # every sample was generated for, the short batch included
        dump = json.load(open(os.path.join(tmp_path, "test_gen_trajs.json")))
        # a sample is dumped when its generation parses (train.py:249-250 skips the others): ids come from the whole range, the tail included
        assert set(int(k) for k in dump) <= set(range(n)) and m["n"] == len(dump)
        assert m["n"] == 0 or np.isfinite(m["ADE"])
