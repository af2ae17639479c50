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
    `finish()` joins.  Averaging (1/world) is folded into the optimizer's grad_scale.

    wire_dtype=torch.bfloat16: fp32 buffers of at least `wire_min_bytes` cross xGMI as bf16 and are widened back in
    finish() — what the reference's DeepSpeed bf16 engine does with its (bf16) gradients (train.py:92-125), at half the
    bytes: 0.55 instead of 1.1 GB per step in frozen-LLM mode, 13.5 instead of 27 GB with every parameter trained.
    Every rank receives the same reduced values, so replicas stay bit-identical."""

    def __init__(self, group=None, wire_dtype=None, wire_min_bytes=32 << 20):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.stream = None
        self.pending = []          # (work, destination buffer, wire buffer or None)
        self.bytes = 0             # bytes handed to the collective (wire size)
        self.wire_dtype, self.wire_min_bytes = wire_dtype, wire_min_bytes

    def ready(self, name, buf: torch.Tensor):
        if self.world == 1:
            return
        wire = None
        if self.wire_dtype is not None and buf.dtype == torch.float32 and buf.numel() * 4 >= self.wire_min_bytes:
            wire = buf.to(self.wire_dtype)                       # on the current stream, before the side stream picks it up
        t = wire if wire is not None else buf
        self.bytes += t.numel() * t.element_size()
        if t.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((w, buf, wire))

    def finish(self):
        for w, _, _ in self.pending:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for _, buf, wire in self.pending:
            if wire is not None:
                buf.copy_(wire)                                  # widen back into the fp32 main_grad buffer
        self.pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world
