"""CPU, world_size 2 over gloo: the data-parallel pieces (sample sharding, overlapped gradient
sum, 1/world folding) behave as SURVEY.md §8e prescribes.  The GPU model itself cannot run here;
these tests drive egoscaler_amd.dp with CPU tensors, which is the same code path RCCL uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egoscaler_amd.dp import GradSync, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(8, rank, world)
        sync = GradSync()
        # "gradients" whose value depends on the rank's shard; ready() in backward order, packed into ONE bucket
        bufs = {n: torch.full((5, 3), float(sum(range(lo, hi)) + i), dtype=torch.float32) for i, n in enumerate(["lm_head", "norm", "embed"])}
        for n in ["lm_head", "norm", "embed"]:
            sync.ready(n, bufs[n])
        sync.finish()
        st1 = dict(sync.stats)
        # bf16 wire with fp32 accumulation: a flat "layer" block, a packed bucket of loose tensors, a small fp32 bucket
        s2 = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1024)
        s2.begin_step()
        layer = torch.full((1000,), float(rank + 1) * 0.5, dtype=torch.float32)             # 1000 is not a multiple of world*8: padded chunks
        s2.ready_flat("layer0", layer)
        noisy = torch.arange(1024, dtype=torch.float32) * (1.0 + 1e-3 * rank) + 0.123
        big = torch.full((64, 16), float(rank + 1) * 0.25, dtype=torch.float32)
        s2.ready("noisy", noisy); s2.ready("big", big)
        s2.flush()
        small = torch.full((4,), float(rank + 1), dtype=torch.float32)
        s2.ready("small", small)
        s2.finish()
        # what "bf16 on the wire, fp32 accumulate" must give: every contribution rounded to bf16 once, summed in fp32 in rank
        # order, the sum rounded to bf16 once for the gather
        contrib = [(torch.arange(1024, dtype=torch.float32) * (1.0 + 1e-3 * r) + 0.123).bfloat16().float() for r in range(world)]
        acc = contrib[0].clone()
        for c in contrib[1:]:
            acc += c
        want_noisy = acc.bfloat16().float()
        wire_ok = bool(torch.equal(layer, torch.full((1000,), 1.5)) and torch.equal(big, torch.full((64, 16), 0.75))
                       and torch.equal(small, torch.full((4,), 3.0)) and big.dtype == torch.float32 and torch.equal(noisy, want_noisy))
        q.put((rank, lo, hi, {n: float(b[0, 0]) for n, b in bufs.items()}, sync.grad_scale, st1, wire_ok, dict(s2.stats), noisy.numpy().copy()))      # numpy: pickled by value (a torch tensor travels by file descriptor and the producer may be gone before the parent reads it: ConnectionResetError, one run in four)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_sync_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 8)]           # contiguous shards of the global batch
    total = sum(range(8))
    for r in res:
        assert r[3] == {"lm_head": total + 0.0, "norm": total + 2.0, "embed": total + 4.0}   # summed over ranks
        assert r[4] == 0.5 and r[5]["buckets"] == 1 and r[5]["collective_calls"] == 1          # three tensors, one fp32 all-reduce
        assert r[6], "bf16-wire buckets must equal the fp32-accumulated sum of bf16-rounded contributions"
        assert r[7]["buckets"] == 3 and r[7]["collective_calls"] == 2 + 2 + 1                  # wire buckets: all-to-all + all-gather
    assert (res[0][8] == res[1][8]).all()                                # replicas hold identical reduced values


def _resident_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1024, resident=True)
        s.begin_step()
        n = 1003                                                       # not a multiple of world * 8: the wire buffer is padded (pad stays zero)
        wire = s.resident_wire("layer0", n, torch.device("cpu"))
        contrib = lambda r: (torch.arange(n, dtype=torch.float32) * (1.0 + 1e-3 * r) + 0.123)       # noqa: E731
        wire[:n] = contrib(rank).bfloat16()                            # what the engine's weight-gradient products write: this rank's bf16 gradient
        s.ready_resident("layer0", wire)
        s.finish()
        acc = contrib(0).bfloat16().float()
        for r in range(1, world):
            acc += contrib(r).bfloat16().float()
        ok = bool(torch.equal(wire[:n].float(), acc.bfloat16().float()) and float(wire[n:].abs().max()) == 0 and wire.dtype == torch.bfloat16)
        same_buffer = s.resident_wire("layer0", n, torch.device("cpu")).data_ptr() == wire.data_ptr()       # persistent per tag
        too_small = s.resident_wire("tiny", 16, torch.device("cpu")) is None                                  # below wire_min_bytes: packed route
        off = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1024).resident_wire("layer0", n, torch.device("cpu")) is None
        q.put((rank, ok, same_buffer, too_small, off, dict(s.stats), wire.float().numpy().copy()))     # numpy: pickled by value
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_resident_exchange_world2_gloo():
    """dp.GradSync(resident=True) on two ranks: the bucket IS its bf16 wire buffer — all-to-all, fp32 rank sum, all-gather in place — and afterwards holds the
    sum over ranks of the bf16 contributions (fp32 accumulation, rounded once), identical on both ranks; the same bits the packed route produces."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_resident_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=90) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3] and r[4], r[:5]
        assert r[5]["buckets"] == 1 and r[5]["collective_calls"] == 2 and r[5]["resident_buckets"] == 1
    assert (res[0][6] == res[1][6]).all()


def test_engine_host_logic_for_stacked_views_and_parameter_groups():
    """Host-only pieces of the round-3 engine: Engine._side_by_side (the stacked [Wq;Wk;Wv] / [Wgate;Wup] operands and gradient blocks are VIEWS of tensors
    that lie back to back in one allocation) and Engine.param_group_of (the order in which the next forward pass reads the parameters)."""
    from egoscaler_amd.engine import Engine
    block = torch.arange(6 * 4, dtype=torch.float32).view(6, 4)
    q, k, v = block[0:2], block[2:4], block[4:6]
    st = Engine._side_by_side([q, k, v])
    assert st is not None and st.shape == (6, 4) and st.data_ptr() == q.data_ptr() and torch.equal(st, block)
    st[3, 1] = -1.0
    assert float(k[1, 1]) == -1.0                                       # a view: writes land in the parameters
    assert Engine._side_by_side([q, v]) is None                         # a gap between them
    assert Engine._side_by_side([q, k.clone()]) is None                 # another allocation
    assert Engine._side_by_side([q, block[2:4, :3]]) is None            # different width / not contiguous
    flat = torch.zeros(64 + 64)
    a, b = flat[0:8].view(2, 4), flat[64:72].view(2, 4)                 # padded to 64 elements each, as the layer's gradient block pads them
    assert Engine._side_by_side([a, b]) is None
    assert Engine.param_group_of("model.layers.17.mlp.up_proj.weight") == 17 and Engine.param_group_of("model.embed_tokens.weight") == "embed"
    assert Engine.param_group_of("lm_head.weight") == "post" and Engine.param_group_of("model.norm.weight") == "post"
    assert Engine.param_group_of("model.point_proj.0.bias") == "pre" and Engine.param_group_of("model.point_backbone.cls_token") == "pre"


def test_shard_range_and_single_process():
    assert shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
    s = GradSync()                      # no process group: world 1, everything is a no-op
    b = torch.ones(4)
    s.ready("x", b)
    s.finish()
    assert s.grad_scale == 1.0 and torch.equal(b, torch.ones(4))


def test_linear_warmup_schedule():
    """train.py:113-116: warm-up from 0 over int(total/5) steps, then linear decay to 0 (pinned against HF's own
    scheduler in tests/test_host_glue.py)."""
    from egoscaler_amd.optim import linear_warmup_lr
    lrs = [linear_warmup_lr(2e-5, s, 100) for s in range(101)]
    assert lrs[0] == 0.0 and abs(lrs[10] - 1e-5) < 1e-12 and abs(lrs[20] - 2e-5) < 1e-12 and abs(lrs[60] - 1e-5) < 1e-12 and lrs[100] == 0.0
