"""GPU: dp.GradSync on the REAL backend (RCCL, `backend="nccl"`), one rank.  RCCL refuses two ranks on one device and the builder's
boxes have one GPU, so this is the most of the RCCL path a one-GPU box can execute: process-group creation with a bound device,
bf16 `all_to_all_single`, `all_gather_into_tensor`, fp32 `all_reduce`, all issued on the side stream the exchange uses, with the
packing / rank-sum / widening kernels around them.  With one rank the sum is the identity up to the bf16 wire rounding."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_grad_sync_collectives_on_rccl_world1():
    import torch.distributed as dist
    from egoscaler_amd.dp import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1 << 16, run_single=True)
        s.begin_step()
        g = torch.Generator(device="cuda").manual_seed(3)
        layer = torch.randn(1_000_003, device="cuda", generator=g)               # flat "decoder layer" block, odd length: padded chunk
        parts = [torch.randn(300, 700, device="cuda", generator=g), torch.randn(4097, device="cuda", generator=g)]
        small = torch.randn(33, device="cuda", generator=g)
        want_layer, want_parts, want_small = layer.bfloat16().float(), [p.bfloat16().float() for p in parts], small.clone()
        s.ready_flat("layer0", layer)
        for i, p in enumerate(parts):
            s.ready(f"p{i}", p)
        s.flush()
        s.ready("small", small)
        s.finish()
        torch.cuda.synchronize()
        assert torch.equal(layer, want_layer)                                     # bf16 on the wire, fp32 accumulate of one contribution
        assert all(torch.equal(a, b) for a, b in zip(parts, want_parts))
        assert torch.equal(small, want_small)                                     # small bucket: fp32 all-reduce, exact
        assert s.stats["buckets"] == 3 and s.stats["collective_calls"] == 2 + 2 + 1
        assert s.backend == "nccl"       # capability comes from the backend's name: RCCL serves all_to_all_single directly, its errors propagate
    finally:
        dist.destroy_process_group()
