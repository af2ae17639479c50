"""Data parallelism for the training step: one process per GPU, samples sharded by rank, gradients
summed with RCCL all-reduce over xGMI on a side stream while the rest of backward runs.

Reference: DeepSpeed ZeRO-1 engine.backward/step (train.py:92-125,183-184) and a DataLoader without
DistributedSampler (train.py:72-82, every rank sees the same batches — fixed here by sharding).
ZeRO-1 optimizer-state sharding is deliberately not reproduced: 288 GB per GPU
holds the replicated fp32 state.  `backend="nccl"` is RCCL on ROCm; the same code runs on gloo/CPU
tensors for the world_size-2 tests.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global batch for `rank`."""
    if global_batch % world:
        raise ValueError("global batch must divide evenly over ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradSync:
    """Sum gradient buffers over ranks.  `ready(name, buf)` may be called as soon as a buffer is
    final; the all-reduce is enqueued on a side stream (GPU) so it overlaps the remaining backward.
    `finish()` joins.  Averaging (1/world) is folded into the optimizer's grad_scale."""

    def __init__(self, group=None, bucket_bytes=256 << 20):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.stream = None
        self.pending = []
        self.bytes = 0

    def ready(self, name, buf: torch.Tensor):
        if self.world == 1:
            return
        self.bytes += buf.numel() * buf.element_size()
        if buf.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    @property
    def grad_scale(self):
        return 1.0 / self.world
