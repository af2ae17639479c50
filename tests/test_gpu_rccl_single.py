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


def test_resident_exchange_with_direct_wire_gradients_on_rccl_world1():
    """`--unfreeze_language_model` under DP with dp.GradSync(resident=True), at a width where the k-major kernel serves the weight gradients:
    the products write bf16 straight into the layer's wire buffer (no packing cast), RCCL runs in place, EgoAdamW reads the wire
    (no widening pass).  One rank on the real backend; against the packed route of the same process: after the first step every decoder
    matrix is bit-identical (their gradients come from ordered sums only; the norm-weight and embedding gradients are fp32 atomic sums whose
    order changes from run to run, so those tensors — and everything after the second step — agree to rounding)."""
    import types
    import torch.distributed as dist
    from egoscaler_amd import synth
    from egoscaler_amd.config import dims_tiny
    from egoscaler_amd.dp import GradSync
    from egoscaler_amd.optim import EgoAdamW
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        dims = dims_tiny()
        dims.lm.hidden_size, dims.lm.num_attention_heads, dims.lm.intermediate_size = 2048, 16, 2816
        B = 8
        toks, masks, Lp = synth.synth_batch(dims, B, text_len=60, num_steps=20, max_traj_token=160)
        pts = torch.stack([synth.synth_cloud(dims, i) for i in range(B)])
        sd = synth.synth_state_dict(dims, 0)
        args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=True, num_bins=dims.tok.num_bins, model_name=None)

        def run(resident):
            m = TrajPointLLMForCausalLM(args, dims, None, device="cuda", dtype=torch.bfloat16)
            m.load_state_dict({k: (v.to(torch.bfloat16) if v.dtype.is_floating_point else v) for k, v in sd.items()})
            m.train()
            opt = EgoAdamW(m, lr=1e-3)
            sync = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1 << 16, run_single=True, resident=resident)
            m.engine.grad_sync = sync
            casts = []
            from egoscaler_amd import ops
            orig = ops.cast

            def spy(x, dtype, out=None):
                casts.append(x.numel())
                return orig(x, dtype, out=out)
            ops.cast = spy
            try:
                losses, w1 = [], None
                for _ in range(2):
                    losses.append(float(m.loss_and_backward(toks.cuda(), masks.cuda(), pts.cuda(), Lp, dims.tok.pad, fps_start=list(range(B)))))
                    sync.finish()
                    red = {n: v.dtype for n, v in m.engine.reduced_grad.items()}
                    opt.step(grad_scale=sync.grad_scale)
                    if w1 is None:
                        w1 = {n: p.detach().clone() for n, p in m.named_parameters()}
            finally:
                ops.cast = orig
            torch.cuda.synchronize()
            return w1, {n: p.detach().clone() for n, p in m.named_parameters()}, losses, dict(sync.stats), red, max(casts)

        w1_a, w_a, l_a, st_a, red_a, big_a = run(False)
        w1_b, w_b, l_b, st_b, red_b, big_b = run(True)
        L, d = dims.lm.num_hidden_layers, dims.lm.hidden_size
        assert not red_a and len(red_b) == 9 * L and all(t == torch.bfloat16 for t in red_b.values())
        assert st_b["resident_buckets"] == L and st_b["buckets"] == st_a["buckets"] and st_b["wire_bytes"] == st_a["wire_bytes"]
        assert big_a >= 3 * d * d            # packed route: the layer blocks go through the packing / widening casts ...
        assert big_b < d * d                 # ... resident route: nothing of a weight's size is cast inside a decoder layer (norm weights only)
        assert abs(l_a[0] - l_b[0]) <= 1e-6 * abs(l_a[0]) and abs(l_a[1] - l_b[1]) <= 1e-4 * abs(l_a[1]) and l_a[1] < l_a[0]
        for n in w1_a:
            if n.startswith("model.layers.") and w1_a[n].dim() == 2:
                assert torch.equal(w1_a[n], w1_b[n]), n
            assert float((w_a[n].float() - w_b[n].float()).abs().max()) <= 2e-2 * float(w_a[n].float().abs().max()) + 1e-6, n
    finally:
        dist.destroy_process_group()
