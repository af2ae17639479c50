#!/usr/bin/env python3
"""Probe: can two ranks on ONE GPU form an RCCL group (so dp.GradSync's all-to-all / all-gather path can run on the real
backend on a one-GPU box)?  Expected to be refused ("Duplicate GPU detected"); bounded by the caller's timeout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        from egoscaler_amd.dp import GradSync
        s = GradSync(wire_dtype=torch.bfloat16, wire_min_bytes=1024)
        x = torch.full((1 << 20,), float(rank + 1), device="cuda")
        s.ready_flat("x", x)
        s.finish()
        torch.cuda.synchronize()
        print(f"rank {rank}: RCCL on a shared GPU worked, sum = {float(x[0])}", flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
