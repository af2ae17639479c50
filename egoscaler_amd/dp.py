"""Data parallelism for the training step: one process per GPU, samples sharded by rank, gradients
summed over ranks on a side stream while the rest of backward runs.

Reference: DeepSpeed ZeRO-1 engine.backward/step (train.py:92-125,183-184: bf16 engine, 5e8-element reduce buckets) and a
DataLoader without DistributedSampler (train.py:72-82, every rank sees the same batches — fixed here by sharding).
ZeRO-1 optimizer-state sharding is deliberately not reproduced: 288 GB per GPU holds the replicated fp32
state.  `backend="nccl"` is RCCL on ROCm; the same code runs on gloo / host tensors for the world_size-2 CPU tests.

Exchange ("bf16 on the wire / fp32 accumulate, bucketed per decoder layer, direct reduce-scatter +
all-gather across all 7 xGMI links rather than a ring"):
  bucket = one flat buffer (a decoder layer's gradients are allocated flat by the engine; loose tensors are packed)
  large buckets :  fp32 -> bf16 wire [W, c]  --all_to_all-->  chunk r of every rank  --egomi_rank_sum (fp32 accumulate,
                   rank order 0..W-1)--> bf16 [c]  --all_gather-->  bf16 [W*c]  --> widened back into the fp32 buffer.
                   Each GPU sends (W-1)/W of the bucket once per phase, to all peers at the same time (xGMI is
                   point-to-point: a ring would push the whole bucket through ONE link per direction).
  small buckets :  one fp32 all-reduce.
Every rank ends with bit-identical values (the all-gather distributes one copy of each reduced chunk), so replicas do not
drift.  Averaging (1/world) is folded into the optimizer's grad_scale.

resident=True (EgoAdamW + bf16 decoder layers, `--unfreeze_language_model`): a decoder layer's bucket IS its bf16 wire buffer, kept
per layer.  The engine's weight-gradient products write their bf16 result straight into it (the rounding the packing cast would
apply, applied by the GEMM epilogue), the collectives run in place, and the optimizer reads the rank-summed bf16 gradient from it
(egomi_adamw_g16): no fp32 -> bf16 packing pass and no bf16 -> fp32 widening pass, 12 of the 24 B per parameter the exchange moved
through HBM on every rank.  Values are bit-identical to the packed route.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global batch for `rank`."""
    if global_batch % world:
        raise ValueError("global batch must divide evenly over ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def ragged_shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n items that need not divide over the ranks (the short last batch of a validation pass: the
    reference's val / test DataLoader has no drop_last, train.py:79-82): the first n % world ranks take one item more; a rank may
    get an empty range."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _cast(src, dst):
    """dst <- src (dtype change allowed).  Device tensors go through libegomi's vector cast; host tensors (gloo
    rehearsal of the N>1 logic on CPUs) through torch."""
    if src.is_cuda:
        from . import ops
        ops.cast(src, dst.dtype, out=dst)
    else:
        dst.copy_(src)


def _rank_sum(chunks, out):
    """chunks [W, c] -> out [c], fp32 accumulation in rank order."""
    if chunks.is_cuda:
        from . import ops
        ops.rank_sum(chunks, out)
    else:
        acc = chunks[0].float().clone()
        for w in range(1, chunks.shape[0]):
            acc += chunks[w].float()
        out.copy_(acc)


class GradSync:
    """`ready(name, buf)` hands over a final gradient buffer (packed into the open bucket), `ready_flat(tag, flat)` a
    buffer that already is a bucket (a decoder layer's flat gradient block); `flush()` closes the open bucket; buckets are
    reduced on a side stream as soon as they close; `finish()` joins.  wire_dtype=torch.bfloat16: buckets of at least
    `wire_min_bytes` cross the fabric as bf16 with fp32 accumulation (see the module docstring)."""

    def __init__(self, group=None, wire_dtype=None, wire_min_bytes=32 << 20, bucket_bytes=256 << 20, run_single=False, resident=False, local=False):
        """bucket_bytes: loose tensors are packed until the open bucket reaches this size.  256 MB: in frozen-LLM mode the lm_head
        gradient (0.53 GB, final at the very START of backward) closes its own bucket at once and crosses the fabric under the 32
        layers' backward; with 1 GB it sat in the open bucket until the embedding gradient arrived at the end of backward and
        both were reduced exposed.
        run_single=True: a one-rank group still goes through every collective (tests drive the real RCCL backend that way
        on a one-GPU box; RCCL refuses two ranks on one device)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.skip = self.world == 1 and not (run_single and dist.is_initialized())
        # local=True on ONE rank (no process group): nothing is exchanged, but with resident=True the decoder layers' weight gradients are still
        # produced in wire precision inside per-layer wire buffers and read there by EgoAdamW — the same bf16 gradients a multi-rank job trains on
        # (and the reference's DeepSpeed bf16 engine: bf16 gradients, fp32 masters, train.py:92-104), 8 B per parameter less through HBM per step.
        self.local = bool(local) and self.skip
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stream = None
        self.wire_dtype, self.wire_min_bytes, self.bucket_bytes = wire_dtype, wire_min_bytes, bucket_bytes
        self.open, self.open_bytes = [], 0         # (name, buf) of the bucket being packed
        self.pending = []                          # works of async host collectives
        self._scratch = {}                         # persistent wire / packing buffers by (role, numel, dtype): buckets are processed in stream
                                                   # order on ONE side stream, so buckets of equal size share their scratch
        self.stats = {"buckets": 0, "collective_calls": 0, "wire_bytes": 0}
        self._bucket_id = 0
        # capability, decided ONCE from the backend's name (never by catching an exception around the real exchange: a rank-local
        # failure would send that rank down the other branch while its peers sit in the first collective — ADVICE r2).  RCCL ("nccl")
        # has all-to-all, and so has gloo for host tensors; gloo with DEVICE tensors (two ranks rehearsing on one card) has not and
        # gathers everything instead.  On RCCL every error of a collective propagates.
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.resident = bool(resident) and wire_dtype is not None   # decoder-layer buckets stay in their wire buffers (module docstring)
        self._resident = {}                        # tag -> persistent wire buffer [W * c]
        self.time_exposed = False                  # bench.py: event pair around finish()'s join = what the exchange costs the step
        self._exposed = []

    # ------------------------------------------------------------------ bucket assembly
    def begin_step(self):
        self.stats = {"buckets": 0, "collective_calls": 0, "wire_bytes": 0}
        self._bucket_id = 0

    def ready(self, name, buf: torch.Tensor):
        if self.skip:
            return
        self.open.append((name, buf))
        self.open_bytes += buf.numel() * buf.element_size()
        if self.open_bytes >= self.bucket_bytes:
            self.flush()

    def ready_flat(self, tag, flat: torch.Tensor):
        if self.skip:
            return
        self.flush()
        self._reduce(flat.view(-1), None)

    def resident_wire(self, tag, n, device):
        """The persistent wire buffer [W * c] of bucket `tag` holding n elements (pad zero), or None when the bucket would not travel
        in wire_dtype (too small / resident mode off / nothing to exchange)."""
        if (self.skip and not self.local) or not self.resident or n * 4 < self.wire_min_bytes:
            return None
        W = self.world
        c = -(-n // (W * 8)) * 8
        t = self._resident.get(tag)
        if t is None or t.numel() != W * c or t.device != device:
            t = self._resident[tag] = torch.zeros(W * c, dtype=self.wire_dtype, device=device)
        return t

    def ready_resident(self, tag, wire):
        """`wire` (resident_wire(tag, ...)) holds this rank's gradients in wire precision: exchanged in place; afterwards it holds the sum
        over ranks, for the optimizer to read."""
        if self.skip:                                  # local mode: one rank, the wire buffer already holds the sum
            return
        self.flush()
        self._bucket_id += 1
        W, cuda = self.world, wire.is_cuda
        c = wire.numel() // W
        if cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(self.stream) if cuda else _Null()):
            recv = self._buf("recv", W * c, self.wire_dtype, wire.device)
            red = self._buf("red", c, self.wire_dtype, wire.device)
            self._all_to_all(recv, wire, W, c)
            _rank_sum(recv.view(W, c), red)
            dist.all_gather_into_tensor(wire, red, group=self.group)
        self.stats["collective_calls"] += 2
        self.stats["wire_bytes"] += 2 * (W - 1) * c * wire.element_size()
        self.stats["buckets"] += 1
        self.stats["resident_buckets"] = self.stats.get("resident_buckets", 0) + 1

    def flush(self):
        if self.skip or not self.open:
            return
        bufs = [b for _, b in self.open]
        self.open, self.open_bytes = [], 0
        if len(bufs) == 1 and bufs[0].is_contiguous():
            self._reduce(bufs[0].view(-1), None)
        else:
            self._reduce(None, bufs)

    # ------------------------------------------------------------------ one bucket
    def _buf(self, key, numel, dtype, device):
        k = (key, numel, dtype)
        t = self._scratch.get(k)
        if t is None:
            t = self._scratch[k] = torch.empty(numel, dtype=dtype, device=device)
        return t

    def _reduce(self, flat, parts):
        """flat: a contiguous 1-D fp32 buffer reduced in place, or parts: tensors packed into one bucket and unpacked after."""
        self._bucket_id += 1
        ref = flat if flat is not None else parts[0]
        dev, cuda = ref.device, ref.is_cuda
        n = flat.numel() if flat is not None else sum(p.numel() for p in parts)
        W = self.world
        use_wire = self.wire_dtype is not None and ref.dtype == torch.float32 and n * 4 >= self.wire_min_bytes
        cur = torch.cuda.current_stream() if cuda else None
        if cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(cur)                      # the bucket's gradients are final on the compute stream
        ctx = torch.cuda.stream(self.stream) if cuda else _Null()
        with ctx:
            if use_wire:
                c = -(-n // (W * 8)) * 8                       # per-rank chunk, 16-B aligned in bf16
                wire = self._buf("wire", W * c, self.wire_dtype, dev)
                if W * c > n:
                    wire[n:].zero_()
                self._pack(flat, parts, wire)
                recv = self._buf("recv", W * c, self.wire_dtype, dev)
                red = self._buf("red", c, self.wire_dtype, dev)
                self._all_to_all(recv, wire, W, c)
                _rank_sum(recv.view(W, c), red)
                dist.all_gather_into_tensor(wire, red, group=self.group)
                self.stats["collective_calls"] += 2
                self.stats["wire_bytes"] += 2 * (W - 1) * c * wire.element_size()
                self._unpack(wire, flat, parts)
            else:
                if flat is not None:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    pk = self._buf("pack", n, ref.dtype, dev)
                    self._pack(None, parts, pk)
                    dist.all_reduce(pk, op=dist.ReduceOp.SUM, group=self.group)
                    self._unpack(pk, None, parts)
                self.stats["collective_calls"] += 1
                self.stats["wire_bytes"] += 2 * (W - 1) * n * ref.element_size() // W
        self.stats["buckets"] += 1

    def _all_to_all(self, recv, send, W, c):
        if self.backend != "gloo" or not send.is_cuda:
            dist.all_to_all_single(recv, send, group=self.group)
            return
        full = [torch.empty_like(send) for _ in range(W)]        # gloo + device tensors only: gather everything, keep my chunk column
        dist.all_gather(full, send, group=self.group)
        for w in range(W):
            recv[w * c:(w + 1) * c].copy_(full[w][self.rank * c:(self.rank + 1) * c])

    @staticmethod
    def _pack(flat, parts, dst):
        if flat is not None:
            _cast(flat, dst[:flat.numel()])
            return
        o = 0
        for p in parts:
            _cast(p.reshape(-1), dst[o:o + p.numel()])
            o += p.numel()

    @staticmethod
    def _unpack(src, flat, parts):
        if flat is not None:
            _cast(src[:flat.numel()], flat)
            return
        o = 0
        for p in parts:
            if p.is_contiguous():
                _cast(src[o:o + p.numel()], p.view(-1))
            else:
                p.copy_(src[o:o + p.numel()].view(p.shape))
            o += p.numel()

    def finish(self):
        self.flush()
        if self.stream is not None:
            cur = torch.cuda.current_stream()
            if self.time_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.stream)
                e1.record(cur)
                self._exposed.append((e0, e1))
            else:
                cur.wait_stream(self.stream)

    def exposed_ms(self):
        """Mean time the compute stream spent waiting for the exchange in finish() (time_exposed=True), per step."""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    @property
    def grad_scale(self):
        return 1.0 / self.world


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
