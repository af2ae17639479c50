#!/usr/bin/env python3
"""bench.py — clips/sec/GPU of the EgoScaler trajectory-generator training step on MI355X.

One step = one pass of the hot path over one batch of synthetic input already resident in HBM:
  A1 un-project 8 RGB-D frames/sample -> A2 pc_norm -> A3-A8 PointBERT -> A9 projector -> A10 splice
  -> A11 32 LLaMA-7B layers -> A12 lm_head + CE on the trajectory span -> full backward -> (N>1: RCCL
  gradient all-reduce, overlapped) -> AdamW step.
Config = BASELINE.json configs[1]: bs=8 per GPU, 8-frame 224x224 clips, 16-token text, bf16.

  python bench.py --gpus N --steps K --warmup W
N>1: either launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the environment), or — from a plain shell —
this process spawns one fresh child per GPU itself BEFORE touching the GPU and waits for them (no re-exec of a process
that initialised HIP).  Every rank asserts dist.get_world_size() == N.
Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant
kernel (the MFMA GEMM) from HIP events recorded in the timed region, and `cpu_baseline` (the CPU
oracle timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)


def flops_per_sample(dims, S, S_traj, frozen_llm=True):
    """Algorithmic FLOPs per sample (SURVEY.md §8d 'Roofline - dense part')."""
    lm, pb = dims.lm, dims.pb
    d, f, L, V = lm.hidden_size, lm.intermediate_size, lm.num_hidden_layers, lm.vocab_size
    G, K, D, P = pb.num_group, pb.group_size, pb.trans_dim, pb.point_token_len
    pointnet = 2 * G * K * (pb.point_dims * pb.pn_c1 + pb.pn_c1 * pb.pn_c2 + 2 * pb.pn_c2 * pb.pn_c3 + pb.pn_c3 * pb.encoder_dims)
    blk = 2 * P * (3 * D * D + D * D + 2 * pb.mlp_ratio * D * D) + 4 * P * P * D
    pointbert = pb.depth * blk + 2 * G * pb.encoder_dims * D
    dims_p = [D] + list(pb.projection_hidden_dim) + [d]
    proj = sum(2 * P * a * b for a, b in zip(dims_p[:-1], dims_p[1:]))
    lin = L * 2 * S * (4 * d * d + 3 * d * f)
    attn = L * 4 * S * S * d // 2
    head = 2 * d * V * S_traj
    fwd_front = pointnet + pointbert + proj
    fwd_llm = lin + attn + head
    bwd = (fwd_llm if frozen_llm else 2 * fwd_llm) + 2 * proj      # dgrad only through a frozen LLM; frozen encoder has no backward
    return {"fwd": fwd_front + fwd_llm, "fwd_bwd": fwd_front + fwd_llm + bwd}


def usable_cores():
    """Host threads this process may actually run on: min(os.cpu_count(), scheduler affinity, cgroup CPU quota); when no quota is
    visible the 1-GPU share of the box (16 CPUs per GPU) caps it.  Measured why: the GPU box reports 256 logical CPUs but
    gives a 1-GPU job a 16-CPU share — 256 oracle threads ran 45x SLOWER than 16 (58.5 s vs 1.3 s for the 1-layer slice)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except Exception:
            pass
    return min(n, quota) if quota else min(n, 16 * max(1, torch.cuda.device_count()))


def cpu_baseline(dims, Lp, threads):
    """Oracle (CPU restatement of the reference) on a bounded sample: ONE clip, full PointBERT + projector +
    splice + lm_head/CE, with a 1-layer and a 4-layer slice of the LLaMA-7B-width stack, fp32, forward+backward (frozen-LLM
    mode), on ALL host cores.  Per-layer time = (t(4)-t(1))/3; whole model = front + t(1) + 31 layers."""
    import copy
    from egoscaler_amd import synth
    from oracle import pointllm as OPL, llama as OL, pointcloud as OPC
    torch.set_num_threads(threads)
    rgb, depth = synth.synth_clip(0, 8, 224, 224)
    f, pp = synth.clip_intrinsics(224)
    t0 = time.time()
    pc = torch.from_numpy(OPC.clip_to_cloud(rgb, depth, pp, f, synth.DEPTH_THRESHOLD, dims.pb.npoints))[None]
    t_front = time.time() - t0
    toks, masks, _ = synth.synth_batch(dims, 1)
    times = {}
    for L in (1, 4):
        dd = copy.deepcopy(dims)
        dd.lm.num_hidden_layers = L
        g = torch.Generator().manual_seed(0)
        sd = {}
        for k, shp in synth.param_shapes(dd):
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                sd[k] = torch.zeros((), dtype=torch.long)
            elif leaf == "running_var" or (leaf == "weight" and len(shp) == 1):
                sd[k] = torch.ones(shp)
            else:
                sd[k] = torch.randn(shp, generator=g) * 0.02
        train = lambda k: not (k.startswith("model.layers.") or k.startswith("model.point_backbone."))
        sd = {k: v.requires_grad_(v.dtype.is_floating_point and train(k)) for k, v in sd.items()}
        t0 = time.time()
        logits = OPL.forward(sd, dd, toks, masks, pc, np.array([0]))
        loss = OL.traj_loss(logits, toks, Lp, dd.tok.pad)
        loss.backward()
        times[L] = time.time() - t0
        del sd, logits, loss
    per_layer = max((times[4] - times[1]) / 3.0, 1e-6)
    total = t_front + times[1] + (dims.lm.num_hidden_layers - 1) * per_layer
    return {"value": 1.0 / total, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": (f"1 clip (8x224x224) fwd+bwd fp32, frozen-LLM mode, oracle on {threads} host threads (every core this job may use: bench.usable_cores(); the box reports {os.cpu_count()} logical CPUs): "
                       f"un-projection+pc_norm {t_front:.2f}s, PointBERT+projector+1 LLaMA-7B-width layer+lm_head/CE {times[1]:.2f}s, 4-layer slice "
                       f"{times[4]:.2f}s -> {per_layer:.2f}s per extra layer; 32-layer time extrapolated linearly = {total:.1f}s")}


class SmiSampler:
    """Engine clock and socket power while the timed region runs (rocm-smi as a child process every ~0.4 s): the dominant
    kernel is power-limited, so a box-to-box difference in ms/step shows up here."""

    def __init__(self, device_index):
        import threading
        self.dev, self.samples, self.on, self.stop = device_index, [], False, False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _read(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            return None
        sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
        pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
        return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None)

    def _run(self):
        while not self.stop:
            if self.on:
                r = self._read()
                if r is not None:
                    self.samples.append(r)
            time.sleep(0.4)

    def summary(self):
        self.stop = True
        self.th.join(6)
        sc = sorted(x[0] for x in self.samples if x[0] is not None)
        pw = sorted(x[1] for x in self.samples if x[1] is not None)
        med = lambda v: v[len(v) // 2] if v else None
        return {"sclk_mhz_median": med(sc), "sclk_mhz_min": sc[0] if sc else None, "socket_power_w_median": med(pw), "samples": len(self.samples),
                "source": "rocm-smi --showclocks --showpower sampled during the timed region"}


def launch_ranks(a):
    """`python bench.py --gpus N` from a plain shell: one fresh child per GPU, started before this process makes any GPU
    call (it never does), environment as torch.distributed.run would set it.  The children are POLLED: as soon as one exits
    non-zero the others are terminated (a rank that died in init_process_group would otherwise leave its peers blocked in the
    rendezvous until somebody's time limit) and the parent exits non-zero at once.  Fresh children only, nothing is re-executed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                if r != 0:
                    rc = abs(r) or 1
                    print(f"bench.py: rank {procs.index(p)} exited with code {r}; stopping the other {len(live)} rank(s)", file=sys.stderr, flush=True)
    for p in live:                                       # only reached with rc != 0: the exact processes this function started
        p.terminate()
    for p in live:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--mode", default="frozen", choices=["frozen", "unfrozen", "pc"],
                    help="frozen: the reference's default flags (model_arch.py:33-51).  unfrozen: --unfreeze_language_model.  pc: --unfreeze_pc_encoder "
                         "(frozen LLM, trainable PointBERT: batch-statistics BatchNorm, DropPath, backward through the point backbone)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer LLaMA layers (result is then marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs run after the headline measurement (config.extra: decode = configs[4], "
                                                             "point branch = configs[3], --mode unfrozen); they never touch `value`")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--gemm-event-steps", type=int, default=2,
                    help="timed steps whose dominant-kernel launches are bracketed by HIP events (evenly spaced, the last one included); "
                         "0 = every step.  An event pair costs ~5 us of stream time: all 260 launches of every step measured +1.3 ms/step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only to rehearse the N>1 code path on one GPU")
    ap.add_argument("--workload", default="train", choices=["train", "decode"],
                    help="train: BASELINE.json configs[1], the headline (default).  decode: configs[4], bs=256 greedy decode of 32 steps in one hipGraph "
                         "(tokens/s + HBM roofline; 1 GPU)")
    ap.add_argument("--force-dist", action="store_true", help="N=1 only: create a one-rank RCCL group and run the gradient exchange through it anyway "
                                                              "(every collective of the N>1 step on the real backend; the line is marked rehearsal)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher self-test: ranks rendezvous (gloo, host tensors), check the world size and exit without touching the GPU")
    a = ap.parse_args()

    if a.workload == "decode":
        if a.gpus != 1:
            raise SystemExit("--workload decode is a single-GPU configuration (BASELINE.json configs[4])")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_decode
        print(json.dumps(bench_decode.run(batch=256, steps=32)), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))                    # parent: spawns the ranks, never touches the GPU itself
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("EGOMI_BENCH_FAIL_RANK") == str(rank):      # launcher self-test (tests/test_bench_launcher.py): this rank dies at start-up
        sys.exit(3)
    import datetime
    rdv_timeout = datetime.timedelta(seconds=int(os.environ.get("EGOMI_BENCH_RDV_TIMEOUT", "120")))
    if a.dry_launch:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", timeout=rdv_timeout)
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            assert dist.get_world_size() == a.gpus and float(t) == world * (world - 1) / 2
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "local_ranks": "0..%d" % (world - 1)}), flush=True)
        return
    ndev = torch.cuda.device_count()                         # counting devices does not initialise HIP
    if a.backend == "nccl" and world > ndev:
        raise SystemExit(f"--gpus {a.gpus} over RCCL needs {a.gpus} visible GPUs, found {ndev} (use --backend gloo to rehearse on fewer)")
    local = local % max(1, ndev)                             # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=rdv_timeout)    # RCCL over xGMI; a missing peer is an error after 120 s, not a hang
        else:
            dist.init_process_group("gloo", timeout=rdv_timeout)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
    elif a.force_dist:
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from egoscaler_amd import ops, synth
    from egoscaler_amd.config import dims_7b
    from egoscaler_amd.dp import GradSync
    from egoscaler_amd.optim import EgoAdamW
    from egoscaler_amd.pointllm import TrajPointLLMForCausalLM

    dims = dims_7b()
    if a.layers is not None:
        dims.lm.num_hidden_layers = a.layers
    B, T, H, W = a.batch, 8, 224, 224

    def build(mode):
        margs = types.SimpleNamespace(unfreeze_pc_encoder=(mode == "pc"), unfreeze_language_model=(mode == "unfrozen"), num_bins=256, model_name=None)
        model = TrajPointLLMForCausalLM(margs, dims, None, device=dev, dtype=torch.bfloat16)
        g = torch.Generator(device=dev).manual_seed(1234)          # same weights on every rank
        with torch.no_grad():
            for n, p in list(model.named_parameters()) + list(model.named_buffers()):
                leaf = n.rsplit(".", 1)[-1]
                if leaf == "num_batches_tracked":
                    continue
                if leaf == "running_var" or (leaf == "weight" and p.dim() == 1):
                    p.fill_(1.0)
                elif leaf == "running_mean":
                    p.zero_()
                else:
                    fan_in = p[0].numel() if p.dim() > 1 else p.numel()
                    std = 0.02 if fan_in >= 1024 else min(0.35, fan_in ** -0.5)
                    tmp = torch.empty(p.shape, dtype=torch.float32, device=dev).normal_(0, std, generator=g) if p.numel() < (1 << 28) else None
                    if tmp is not None:
                        p.copy_(tmp)
                    else:
                        for r0 in range(0, p.shape[0], 4096):
                            blk = p[r0:r0 + 4096]
                            blk.copy_(torch.empty(blk.shape, dtype=torch.float32, device=dev).normal_(0, std, generator=g))
        model.engine.prepared = False
        model.train()
        opt = EgoAdamW(model, lr=2e-5)
        sync = GradSync(wire_dtype=torch.bfloat16, run_single=a.force_dist, resident=True) if (world > 1 or a.force_dist) else None      # large fp32 gradient buffers cross xGMI as bf16
        if sync is None and mode == "unfrozen" and os.environ.get("EGOMI_BF16_GRADS", "1") != "0":
            sync = GradSync(wire_dtype=torch.bfloat16, resident=True, local=True)     # one rank: the decoder layers' weight gradients in bf16 wire buffers, as a DP job (and the reference's bf16 engine) has them
        if sync is not None:
            sync.time_exposed = True
        model.engine.grad_sync = sync
        return model, opt, sync

    # ---- synthetic batch, resident in HBM before the timed region (rank r gets samples r*B .. r*B+B-1)
    clips = [synth.synth_clip(rank * B + i, T, H, W) for i in range(B)]
    rgb = torch.from_numpy(np.stack([c[0] for c in clips])).to(dev)
    depth = torch.from_numpy(np.stack([c[1] for c in clips])).to(dev)
    toks, masks, Lp = synth.synth_batch(dims, B, text_len=16, num_steps=20, max_traj_token=160, first_id=rank * B)
    toks, masks = toks.to(dev), masks.to(dev)
    S = toks.shape[1]
    fx, pp = synth.clip_intrinsics(H)
    fps_start = torch.zeros(B, dtype=torch.int32, device=dev)
    N = dims.pb.npoints

    # AdamW under the next step's forward pass (EgoAdamW.step(overlap=True)) where most parameters train: unfrozen 226.8 -> 224.1 ms.  With the frozen LLM
    # (1.4 ms of AdamW) it measured -0.35 ... +0.5 ms — the update's waves slow the latency-bound point branch it lands on (FPS 0.57 -> 1.18 ms) — so the
    # headline mode keeps the plain step.  EGOMI_OPT_OVERLAP=0 / 1 forces either (A/B runs)
    OVERLAP_ENV = os.environ.get("EGOMI_OPT_OVERLAP")                               # default: overlapped where decoder layers train (also the `unfrozen` leg of config.extra)
    EARLY_ENV = os.environ.get("EGOMI_OPT_EARLY", "1") != "0"                      # EgoAdamW.arm(): per-layer updates as soon as a layer's gradients are final

    def barrier():
        if world > 1:
            dist.barrier()

    def measure(model, opt, sync, steps, warmup, gemm_event_steps, smi):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides (the bench contract)."""
        OVERLAP_OPT = (OVERLAP_ENV != "0") if OVERLAP_ENV is not None else model.engine.any_layer_trainable
        EARLY_OPT = OVERLAP_OPT and EARLY_ENV

        def step(check=False):
            pts, col, cnt = ops.unproject_gather(rgb, depth, pp, fx, fx, synth.DEPTH_THRESHOLD, n_out=N)        # A1
            if check and int(cnt.min()) < N:
                raise RuntimeError("synthetic clip has too few valid pixels")
            pc = ops.pc_norm(pts, col)                                                                            # A2
            if EARLY_OPT and (sync is None or sync.local):
                opt.arm(grad_scale=1.0)                                       # trainable decoder layers: AdamW under the backward pass of the layers below (one rank)
            loss = model.loss_and_backward(toks, masks, pc, Lp, dims.tok.pad, fps_start=fps_start)                # A3-A15
            if sync is not None:
                sync.finish()
            opt.step(grad_scale=sync.grad_scale if sync is not None else 1.0, overlap=OVERLAP_OPT)   # the update runs under the next step's forward pass
            return loss
        for i in range(warmup):
            step(check=(i == 0))
        torch.cuda.synchronize()
        if sync is not None:
            sync._exposed.clear()
        prof = None
        if gemm_event_steps is not None:
            prof = ops.GemmProfiler(min_flops=0, kernel_ids=(2,))          # the dominant kernel only: the 8-phase GEMM, both tile forms
            ops.PROFILER = prof
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        barrier()
        torch.cuda.synchronize()
        if smi is not None:
            smi.on = True
        t0 = time.perf_counter()
        marks[0].record()
        n_ev = steps if (prof is None or gemm_event_steps <= 0 or gemm_event_steps >= steps) else gemm_event_steps
        ev_steps = {steps - 1 - (j * steps) // n_ev for j in range(n_ev)} if prof is not None else set()
        loss = None
        for i in range(steps):
            if prof is not None:
                prof.enabled = i in ev_steps
            loss = step()
            marks[i + 1].record()                                 # stream-ordered step boundaries: no host sync inside the timed region
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if smi is not None:
            smi.on = False
        ops.PROFILER = None
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax), per_step, float(loss), prof, ev_steps

    model, opt, sync = build(a.mode)
    smi = SmiSampler(local) if rank == 0 else None
    dt, per_step, loss, prof, ev_steps = measure(model, opt, sync, a.steps, a.warmup, None if a.no_gemm_events else a.gemm_event_steps, smi)
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    clips_total = a.steps * B * world
    value = clips_total / dt / world            # clips/sec/GPU ... see `value` note below

    fl = flops_per_sample(dims, S, S - Lp, frozen_llm=(a.mode != "unfrozen"))
    roof = None
    if prof is not None:
        sm = prof.summary()
        ach = sm["flops"] / (sm["ms"] * 1e-3) / 1e12 if sm["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        import glob
        pjs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_gemm.json")))      # the newest round's PMC record
        pj = pjs[-1] if pjs else ""
        if pj and os.path.exists(pj):
            try:
                rec = json.load(open(pj))
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = ("NOT measured in this run: read from profiles/%s = separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                               "passes over `bench.py --steps 1 --warmup 1` (%s)" % (os.path.basename(pj), rec.get("recorded", "round and commit in profiles/README.md")))
            except Exception:
                traffic = None
        roof = {"bound": "mfma", "kernel": "the 8-phase bf16 GEMM = gemm_nt_bf16_8phase_kernel (256x256 tiles) + gemm_nt_bf16_tall_kernel (its 352x256 form; egomi_gemm_kernel_id 2: "
                                           "in rocprofv3's kernel stats the two rows together; a launch here = one egomi_gemm product, which is two kernels back to back where the library splits a product's columns between the two forms) "
                                           "(every product of %d of the %d timed steps; HIP events recorded by the library on the launch "
                                           "stream directly around the kernel, inside the timed region: egomi_gemm_time_next.  avg_call_ms is the whole egomi_gemm call, "
                                           "i.e. plus splitk_reduce_kernel where the tail rows are K-sliced)" % (len(ev_steps), a.steps),
                "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": sm["launches"] // max(1, len(ev_steps)), "launches_timed": sm["launches"], "steps_timed": sorted(ev_steps),
                "avg_launch_ms": round(sm["ms"] / max(1, sm["launches"]), 4), "avg_call_ms": round(sm["call_ms"] / max(1, sm["launches"]), 4),
                "gemm_share_of_step": round(sm["ms"] / max(1, len(ev_steps)) / (dt / a.steps * 1e3), 3)}
    gs = None
    if sync is not None:
        gs = dict(sync.stats)
        gs["exposed_ms_per_step"] = None if sync.exposed_ms() is None else round(sync.exposed_ms(), 3)     # compute stream waiting in GradSync.finish()
        gs["exposed_is"] = "mean HIP-event time the compute stream waits for the exchange's side stream before AdamW (last bucket's tail)"
        if sync.local:
            gs["local"] = "one rank, nothing exchanged: the decoder layers' weight gradients are produced in bf16 wire buffers and read there by AdamW (GradSync(local=True))"
    out = {
        "metric": "clips/sec/GPU (8-frame 224^2, 16-token text) fwd+bwd",
        "value": round(clips_total / dt, 4), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "ms_per_step_median": round(ms_median, 3), "ms_per_step_min_max": [round(per_step[0], 3), round(per_step[-1], 3)],
        "value_is": "whole-job clips/s over all n_gpus (bench contract); the per-GPU figure the metric names is config.per_gpu_clips_per_s",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[1]: bs=8/GPU, 8-frame 224x224 RGB-D clip -> 8192-pt cloud, 16-token text, S=%d, PointBERT-v1.2 + LLaMA-7B shapes, "
                               "fwd+bwd+AdamW, %s-LLM mode (reference default flags)" % (S, a.mode),
                   "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}", "backend": ("rccl" if a.backend == "nccl" else "gloo") if world > 1 else None,
                   "world_size_checked": (dist.get_world_size() if world > 1 else 1), "per_gpu_clips_per_s": round(value, 4),
                   "grad_sync": gs,
                   "algorithmic_tflop_per_clip": round(fl["fwd_bwd"] / 1e12, 3),
                   "model_tflops_per_gpu": round(fl["fwd_bwd"] * value / 1e12, 2), "loss": round(loss, 4),
                   "valid": a.layers is None and B == 8, "rehearsal_one_rank_rccl_group": bool(a.force_dist)},
    }
    if roof is not None:
        out["roofline"] = roof
    if rank == 0:
        out["clocks"] = smi.summary() if smi is not None else None

    # ---- the other single-GPU configurations and the other training mode, AFTER the headline's timed region, in this process (VERDICT r2
    # weak #3: they used to be builder-measured only).  Nothing here touches `value`.  Skipped with --no-extras, for N > 1 and debug sizes.
    if world == 1 and not a.no_extras and not a.force_dist and a.layers is None and B == 8 and a.mode == "frozen":
        extra = {}
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        t_x = time.perf_counter()
        try:
            import bench_decode
            import bench_pointbranch
            model.engine.grad_sync = None
            del opt
            model.eval()
            model.engine.ws.bufs.clear()
            torch.cuda.empty_cache()
            d = bench_decode.run(batch=256, steps=32, model=model)              # configs[4] on the very weights the headline trained
            extra["decode"] = {"workload": "configs[4]: bs=256, prompt 540, 32 greedy steps in ONE hipGraph, bf16", "ms_per_step": d["ms_per_step"],
                               "tokens_per_s": d["value"], "algorithmic_TBps": round(d["roofline"]["achieved"] / 1e3, 3), "frac_of_8TBps": d["roofline"]["frac"],
                               "algorithmic_GB": d["roofline"]["algorithmic_GB"], "prefill_s": d["prefill_s"], "deterministic_replay": d["deterministic_replay"],
                               "formula": "bytes = 31 steps x (2 B x LLM params [layers + norm + lm_head]) + sum_t B x (S0 + t + 1) x 2 x d x 2 B x L of K/V rows (SURVEY.md §8d); "
                                          "time = HIP events around graph replay, mean of 3; tokens/s = B x 32 / time"}
            torch.cuda.empty_cache()
            pbr = bench_pointbranch.run(model=model, with_n4=False)["rows"]    # configs[3] point branch on the same (frozen) PointBERT
            extra["pointbranch"] = {"workload": "configs[3]: B=8, 16-frame 448x448 RGB-D -> 8192-pt clouds + point branch, bf16",
                                    "a1_us": round(pbr["A1_unproject_subsample"]["ms"] * 1e3, 1), "a1_TBps": round(pbr["A1_unproject_subsample"]["GBps"] / 1e3, 3),
                                    "a1_frac_of_8TBps": round(pbr["A1_unproject_subsample"]["GBps"] / 8000.0, 4),
                                    "fps_ms": pbr["A3_fps"]["ms"], "knn_ms": pbr["A4_A5_knn_group"]["ms"], "pointnet_ms": pbr["A6_pointnet"]["ms"],
                                    "pointnet_TFLOPs": pbr["A6_pointnet"]["TFLOPs"], "backbone_ms": pbr["A3_A8_point_backbone_total"]["ms"],
                                    "formula": "A1 bytes = 7 B x pixels read once + 36 B x selected points written (SURVEY.md §8d); times = HIP events, mean of 10 calls "
                                               "(torch.empty of the outputs included)"}
            del model
            torch.cuda.empty_cache()
            mu, ou, su = build("unfrozen")                                       # --unfreeze_language_model: all 6.7 B parameters trained
            dtu, psu, lossu, _, _ = measure(mu, ou, su, 10, 3, None, None)
            flu = flops_per_sample(dims, S, S - Lp, frozen_llm=False)
            cu = 10 * B / dtu
            extra["unfrozen"] = {"workload": "configs[1] with --unfreeze_language_model (every LLM weight trained: wgrads + 6.7 B-parameter AdamW; overlapped optimizer step with per-layer updates under the backward pass, bf16 weight gradients in per-layer wire buffers as on the DP wire — GradSync(local=True); EGOMI_BF16_GRADS=0 / EGOMI_OPT_EARLY=0 switch them off)", "steps": 10, "warmup": 3,
                                 "ms_per_step": round(dtu / 10 * 1e3, 3), "clips_per_s": round(cu, 3), "algorithmic_tflop_per_clip": round(flu["fwd_bwd"] / 1e12, 3),
                                 "frac_of_peak": round(flu["fwd_bwd"] * cu / 1e12 / PEAK_BF16_TFLOPS, 4), "loss": round(lossu, 4),
                                 "formula": "frac_of_peak = algorithmic fwd+bwd FLOP per clip (3x forward for the LLM, SURVEY.md §8d) x clips/s / 2.5 PFLOP/s (whole step, not one kernel)"}
            del mu, ou, su
            torch.cuda.empty_cache()
            mp, op_, sp = build("pc")                                            # --unfreeze_pc_encoder: point backbone trained (train-mode BatchNorm, DropPath), LLM frozen
            dtp, _, lossp, _, _ = measure(mp, op_, sp, 10, 3, None, None)
            extra["pc"] = {"workload": "configs[1] with --unfreeze_pc_encoder (point backbone in train() mode and trained, LLM frozen: model_arch.py:33-36)", "steps": 10, "warmup": 3,
                           "ms_per_step": round(dtp / 10 * 1e3, 3), "clips_per_s": round(10 * B / dtp, 3), "loss": round(lossp, 4)}
            del mp, op_, sp
            torch.cuda.empty_cache()
        except Exception as e:                                                   # an extra leg never takes the headline down
            extra["error"] = f"{type(e).__name__}: {e}"
        extra["seconds"] = round(time.perf_counter() - t_x, 1)
        out["config"]["extra"] = extra
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dims_7b(), Lp, usable_cores())
            except Exception as e:      # the oracle is a reported baseline; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1 or a.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
