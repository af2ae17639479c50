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
        # "gradients" whose value depends on the rank's shard; ready() in backward order
        bufs = {n: torch.full((5, 3), float(sum(range(lo, hi)) + i), dtype=torch.float32) for i, n in enumerate(["lm_head", "norm", "embed"])}
        for n in ["lm_head", "norm", "embed"]:
            sync.ready(n, bufs[n])
        sync.finish()
        # bf16 wire for large fp32 buffers: same sums (these values are exact in bf16), half the bytes, small ones untouched
        s2 = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1024)
        big = torch.full((64, 16), float(rank + 1) * 0.5, dtype=torch.float32)
        small = torch.full((4,), float(rank + 1), dtype=torch.float32)
        noisy = torch.arange(1024, dtype=torch.float32) * (1.0 + 1e-3 * rank) + 0.123
        s2.ready("big", big); s2.ready("small", small); s2.ready("noisy", noisy)
        s2.finish()
        exact = sum(torch.arange(1024, dtype=torch.float32) * (1.0 + 1e-3 * r) + 0.123 for r in range(world))
        wire_ok = bool(torch.equal(big, torch.full((64, 16), 1.5)) and torch.equal(small, torch.full((4,), 3.0))
                       and big.dtype == torch.float32 and float(((noisy - exact).abs() / exact.abs()).max()) < 2 ** -6)
        q.put((rank, lo, hi, {n: float(b[0, 0]) for n, b in bufs.items()}, sync.grad_scale, sync.bytes, wire_ok, s2.bytes, noisy.clone()))
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
        assert r[4] == 0.5 and r[5] == 3 * 5 * 3 * 4
        assert r[6] and r[7] == 64 * 16 * 2 + 4 * 4 + 1024 * 2          # big and noisy travel as bf16, small stays fp32
    assert torch.equal(res[0][8], res[1][8])                             # replicas hold identical reduced values


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
    from egoscaler_amd.optim import linear_warmup_lr
    lrs = [linear_warmup_lr(2e-5, s, 100) for s in range(100)]
    assert abs(lrs[0] - 1e-6) < 1e-12 and abs(lrs[19] - 2e-5) < 1e-12 and lrs[50] == 2e-5
