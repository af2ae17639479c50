"""GPU, two ranks on one MI355X over gloo: the DRIVER's multi-rank control flow — `driver.train` for two epochs with a
short last training step (DistributedSampler-style wrap-around, so both ranks run the same micro-batches), a validation set whose last batch
is smaller than the world (rank 1's share is EMPTY: it skips the generation and still joins the metrics all-reduce), rank-0-only
checkpointing followed by the next epoch's collectives, then `driver.evaluate` with its `all_gather_object` of the dump.  Match
train.py:207-308, evaluate.py:104-170.  RCCL needs one GPU per rank (tools/debug/rccl_same_gpu_probe.py), which a one-GPU box cannot give:
gloo drives the same code; the test's time limit is what catches a rank left waiting for a peer that skipped a collective."""
import json
import os
import socket
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, unfreeze, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from egoscaler_amd import driver, synth
        from egoscaler_amd.config import dims_tiny
        from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
        dims = dims_tiny()
        margs = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=unfreeze, num_bins=dims.tok.num_bins, model_name=None)
        m = TrajPointLLMForCausalLM(margs, dims, None, device="cuda", dtype=torch.bfloat16)
        m.load_state_dict({k: (v.to(torch.bfloat16) if v.dtype.is_floating_point else v) for k, v in synth.synth_state_dict(dims, 0).items()})

        class Shifted(driver.SyntheticTrajData):                       # image ids that are not the dataset indices
            def batch(self, idx, device, max_traj_token=160):
                b = super().batch(idx, device, max_traj_token)
                b["image_ids"] = b["image_ids"] + 100
                return b
        train = Shifted(dims, 5, num_steps=4, text_len=8)              # bs 4: one full step + one step of ONE sample (wrapped to two)
        val = Shifted(dims, 5, num_steps=4, text_len=8, seed=977)      # bs 4: one full batch + a last batch of one sample: rank 1's share is empty
        a = types.SimpleNamespace(bs=4, grad_accum_steps=1, lr_llm=1e-3, epochs=2, max_traj_token=48, num_steps=4, out_dir=out_dir,
                                  checkpoint_dir=out_dir, val_sample=False, resume=False)
        hist = driver.train(a, m, train, val, torch.device("cuda"), log=lambda s: None)
        w = {n: p.detach().float().cpu().numpy() for n, p in m.named_parameters() if n in ("lm_head.weight", "model.point_proj.0.weight", "model.norm.weight")}
        metrics = driver.evaluate(a, m, val, "val", torch.device("cuda"))
        q.put((rank, hist, metrics, w))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
@pytest.mark.parametrize("unfreeze", [False, True])
def test_driver_train_and_evaluate_with_two_ranks(tmp_path, unfreeze):
    import numpy as np
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out_dir = str(tmp_path / "run")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, out_dir, unfreeze, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=360) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, h0, m0, w0), (_, h1, m1, w1) = res
    assert len(h0) == 2 and h0 == h1, (h0, h1)                           # identical records on both ranks: losses, metrics, steps, rates
    assert [r["global_step"] for r in h0] == [2, 4]                      # two optimizer steps per epoch: the short last step is kept
    assert h0[0]["n"] == 5 and m0 == m1 and m0["n"] == 5                 # every validation sample counted once, on whichever rank it ran
    assert all(np.isfinite(r["train_loss"]) and np.isfinite(r["ADE"]) for r in h0)
    for n in w0:
        assert np.array_equal(w0[n], w1[n]), n                           # the replicas did not drift
    dump = json.load(open(os.path.join(out_dir, "val_gen_trajs.json")))
    assert sorted(int(k) for k in dump) == [100, 101, 102, 103, 104]     # every image id once, gathered from both ranks
    assert os.path.exists(os.path.join(out_dir, "latest_model.pt")) and os.path.exists(os.path.join(out_dir, "best_model_ade.pt"))
