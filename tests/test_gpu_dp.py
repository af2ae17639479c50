"""GPU, two ranks on one MI355X over gloo: the data-parallel step (sample shard per rank, dp.GradSync all-reduce overlapped
with backward, 1/world folded into EgoAdamW) reproduces the single-process gradient of the whole batch and keeps the
replicas bit-identical.  RCCL needs one GPU per rank, which the one-GPU box cannot give; gloo drives the same dp code."""
import os
import socket
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dims, unfreeze):
    from egoscaler_amd import synth
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=unfreeze, num_bins=dims.tok.num_bins, model_name=None)
    m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.bfloat16)
    sd = synth.synth_state_dict(dims, 0)
    m.load_state_dict({k: (v.to(torch.bfloat16) if v.dtype.is_floating_point else v) for k, v in sd.items()})
    m.train()
    return m


def _dims():
    from egoscaler_amd.config import dims_tiny
    d = dims_tiny()
    d.lm.hidden_size, d.lm.num_attention_heads, d.lm.intermediate_size = 256, 2, 512
    return d


def _batch(dims, n):
    from egoscaler_amd import synth
    toks, masks, Lp = synth.synth_batch(dims, n, text_len=8, num_steps=4, max_traj_token=40)
    pts = torch.stack([synth.synth_cloud(dims, i) for i in range(n)])
    return toks, masks, Lp, pts, [0, 17, 3, 9][:n]


def _worker(rank, world, port, wire, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from egoscaler_amd.dp import GradSync, shard_range
        from egoscaler_amd.optim import EgoAdamW
        dims = _dims()
        m = _build(dims, True)
        opt = EgoAdamW(m, lr=1e-3)
        sync = GradSync(wire_dtype=torch.bfloat16 if wire else None, wire_min_bytes=1 << 16, resident=(wire == "resident"))
        m.engine.grad_sync = sync
        toks, masks, Lp, pts, start = _batch(dims, 4)
        lo, hi = shard_range(4, rank, world)
        loss = m.loss_and_backward(toks[lo:hi].cuda(), masks[lo:hi].cuda(), pts[lo:hi].cuda(), Lp, dims.tok.pad, fps_start=start[lo:hi])
        sync.finish()
        names = ["lm_head.weight", "model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
                 "model.point_proj.0.weight", "model.norm.weight"]
        # (resident exchange: the decoder layers' reduced gradients live in their bf16 wire buffers, engine.reduced_grad)
        grads = {n: (m.engine.reduced_grad.get(n, m.engine.main_grad[n]).float() * sync.grad_scale).cpu().numpy() for n in names}      # numpy: pickled by value
        opt.step(grad_scale=sync.grad_scale)
        w = {n: dict(m.named_parameters())[n].detach().float().cpu().numpy() for n in names}
        q.put((rank, float(loss), grads, w, dict(sync.stats)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def _two_ranks(wire):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, wire, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
def test_resident_exchange_equals_packed_exchange_bitwise():
    """dp.GradSync(resident=True): the decoder layers' buckets are exchanged inside their bf16 wire buffers and read there by the optimizer
    (egomi_adamw_g16) — the same reduced gradients and the same weights after the step as the packed route, bit for bit, on both ranks."""
    a, b = _two_ranks(True), _two_ranks("resident")
    for r in range(2):
        for n in a[r][2]:
            assert np.array_equal(a[r][2][n], b[r][2][n]), n
            assert np.array_equal(a[r][3][n], b[r][3][n]), n
    assert b[0][4].get("resident_buckets") == _dims().lm.num_hidden_layers and a[0][4]["wire_bytes"] == b[0][4]["wire_bytes"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wire", [False, True])
def test_two_ranks_match_single_process_full_batch(wire):
    res = _two_ranks(wire)
    # single process, whole batch
    dims = _dims()
    m = _build(dims, True)
    toks, masks, Lp, pts, start = _batch(dims, 4)
    loss = m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=start)
    assert abs(0.5 * (res[0][1] + res[1][1]) - float(loss)) < 2e-2 * abs(float(loss))        # equal token counts per shard
    for n, g0 in res[0][2].items():
        ref = m.engine.main_grad[n].float().cpu().numpy()
        tol = (3e-2 if wire else 2e-2) * float(np.abs(ref).max()) + 1e-6
        assert float(np.abs(g0 - ref).max()) <= tol, n                                         # DP mean gradient == full-batch gradient
        assert np.array_equal(g0, res[1][2][n]), n                                            # both ranks hold the same reduced gradient
        assert np.array_equal(res[0][3][n], res[1][3][n]), n                                  # ... and the same weights after the step
    nparam = sum(p.numel() for n, p in m.named_parameters() if n in m.engine.trainable)
    st = res[0][4]
    assert st["buckets"] == 1 + dims.lm.num_hidden_layers + 1, st              # lm_head, one per decoder layer (reverse order), the rest
    if wire:
        assert st["wire_bytes"] < 0.75 * nparam * 4                               # bf16 on the wire (2 phases x (W-1)/W x 2 B)
