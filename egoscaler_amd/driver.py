"""Train / evaluate drivers on RCCL (SURVEY.md §8f rows N1 + N2).

Mirrors egoscaler/models/pointllm/train.py:39-310 and evaluate.py:70-170 without DeepSpeed / wandb:
  * same flags (argparse names of train.py:312-347), AdamW + linear warm-up over 1/5 of the steps
    (train.py:107-117), loss on the trajectory span (train.py:174-181), validation by generation +
    ADE/FDE (train.py:207-264), checkpoints `latest_model.pt` / `best_model_ade.pt` with the
    reference's dict keys (train.py:287-308), `--resume` (train.py:138-150),
    `{split}_gen_trajs.json` dump (evaluate.py:167-170)
  * one process per GPU, batch sharded by rank (the reference has no DistributedSampler,
    train.py:72-82), gradients all-reduced by dp.GradSync overlapped with backward.
The batch dict is the collate_fn contract of dataset.py:150-194 ('pcrgbs', 'prompts', 'prompt_masks',
'tokens', 'attention_masks', 'trajectories', 'trajectory_masks', 'max_abs', 'image_ids'), assembled on
the device (`build_batch`).  The release has no dataset (`__getitem__` is missing), so
the only data source wired here is the seeded synthetic one of SURVEY.md §8d.
"""
import argparse
import json
import os
import types

import numpy as np
import torch
import torch.distributed as dist

from . import ops, synth, traj as T
from .dp import GradSync, shard_range
from .optim import EgoAdamW, linear_warmup_lr


def build_batch(dims, desc_ids, trajs, pcrgbs, max_traj_token=160, image_ids=None, steps=None):
    """desc_ids i64 [B,Ld] (description tokens), trajs f32 [B,T,6] in [-1,1], pcrgbs f32 [B,N,6] (device).
    Sequence: [bos] desc_a <point_start> <point_patch>xP <point_end> desc_b | <ts> (p*6 <tsep>)*T <te> eos pad
    (SURVEY.md §8d); prompts run up to the first <tsep> of sample 0 (dataset.py:180-182)."""
    tok, P = dims.tok, dims.pb.point_token_len
    dev = pcrgbs.device
    B, Ld = desc_ids.shape
    a, b = desc_ids[:, :Ld // 2], desc_ids[:, Ld // 2:]
    col = lambda v: torch.full((B, 1), v, dtype=torch.int64, device=dev)
    head = torch.cat([col(tok.bos), a, col(tok.point_start), torch.full((B, P), tok.point_patch, dtype=torch.int64, device=dev),
                      col(tok.point_end), b], 1)
    tt, tm = T.tokenize_batch(trajs, tok, max_traj_token, steps=steps)
    tokens = torch.cat([head, tt], 1)
    masks = torch.cat([torch.ones_like(head, dtype=torch.bool), tm], 1)
    pos = int((tokens[0] == tok.tsep).nonzero()[0, 0])
    return {"image_ids": image_ids if image_ids is not None else torch.arange(B, device=dev), "pcrgbs": pcrgbs,
            "prompts": tokens[:, :pos + 1], "prompt_masks": masks[:, :pos + 1], "tokens": tokens, "attention_masks": masks,
            "trajectories": trajs, "trajectory_masks": tm, "max_abs": torch.ones(B, 6, device=dev)}


class SyntheticTrajData:
    """Seeded synthetic samples: RGB-D clip -> un-projection -> 8192-point cloud (A1, A2 on the device),
    uniform description ids, smooth random trajectories."""

    def __init__(self, dims, n_samples, frames=2, size=64, text_len=16, num_steps=20, seed=42):
        self.dims, self.n, self.frames, self.size, self.text_len, self.num_steps, self.seed = dims, n_samples, frames, size, text_len, num_steps, seed

    def __len__(self):
        return self.n

    def batch(self, idx, device, max_traj_token=160):
        dims, H = self.dims, self.size
        clips = [synth.synth_clip(int(i), self.frames, H, H, self.seed) for i in idx]
        rgb = torch.from_numpy(np.stack([c[0] for c in clips])).to(device)
        depth = torch.from_numpy(np.stack([c[1] for c in clips])).to(device)
        f, pp = synth.clip_intrinsics(H)
        pts, col, cnt = ops.unproject_gather(rgb, depth, pp, f, f, synth.DEPTH_THRESHOLD, n_out=dims.pb.npoints)
        if int(cnt.min()) < dims.pb.npoints:
            raise ValueError("clip has fewer valid pixels than npoints")
        pc = ops.pc_norm(pts, col)
        desc, trs = [], []
        for i in idx:
            g = synth._rng(self.seed + int(i), 0xDA7A)
            desc.append(g.integers(3, min(dims.tok.point_patch, dims.lm.vocab_size), size=self.text_len))
            base = g.uniform(-0.6, 0.6, size=(1, 6))
            walk = np.cumsum(g.normal(0, 0.03, size=(self.num_steps, 6)), 0)
            trs.append(np.clip(base + walk, -1, 1))
        desc = torch.from_numpy(np.stack(desc)).to(device)
        trs = torch.from_numpy(np.stack(trs).astype(np.float32)).to(device)
        return build_batch(dims, desc, trs, pc, max_traj_token, image_ids=torch.as_tensor(list(idx), device=device))


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


@torch.no_grad()
def run_validation(model, data, args, device, max_batches=None):
    """Generation + metrics on the rank's shard (train.py:207-264): prompts up to the first <tsep>,
    greedy decode of the remaining positions, de-tokenise, pad with the last step, ADE / FDE."""
    dims = model.dims
    rank, world = _rank_world()
    model.eval()
    ades, fdes, dump = [], [], {}
    order = list(range(len(data)))
    nb = len(order) // (args.bs * world)
    for bi in range(nb if max_batches is None else min(nb, max_batches)):
        lo, hi = shard_range(args.bs * world, rank, world)
        idx = order[bi * args.bs * world + lo: bi * args.bs * world + hi]
        batch = data.batch(idx, device, args.max_traj_token)
        prompts, tokens = batch["prompts"], batch["tokens"]
        max_new = tokens.shape[1] - prompts.shape[1]
        out = model.generate(input_ids=prompts, attention_mask=batch["prompt_masks"], point_clouds=batch["pcrgbs"], max_length=max_new,
                             do_sample=False, fps_start=torch.zeros(len(idx), dtype=torch.int32, device=device))
        gen_ids = out.sequences[:, prompts.shape[1]:]
        # the prompt holds the first step; prepend its six tokens + <tsep> so step 0 is parsed like the rest
        first = prompts[:, -7:]
        vals, n = T.detokenize_batch(torch.cat([first, gen_ids], 1), dims.tok, args.num_steps + 4)
        gt = batch["trajectories"]
        Tn = gt.shape[1]
        ade, fde = T.metrics_batch(vals[:, :Tn].contiguous(), torch.clamp(n, max=Tn), gt)
        ok = n > 0
        ades += ade[ok].tolist()
        fdes += fde[ok].tolist()
        for j, i in enumerate(idx):
            dump[int(i)] = {"gen_traj": T.denorm(vals[j:j + 1, :max(1, min(int(n[j]), Tn))].cpu().numpy())[0].tolist()}
    model.train()
    return {"ADE": float(np.mean(ades)) if ades else float("nan"), "FDE": float(np.mean(fdes)) if fdes else float("nan"), "n": len(ades)}, dump


def train(args, model, train_data, val_data=None, device="cuda", log=print):
    """train.py:129-310.  Returns the list of per-epoch records."""
    rank, world = _rank_world()
    opt = EgoAdamW(model, lr=args.lr_llm)
    sync = GradSync(wire_dtype=torch.bfloat16 if model.engine.dtype == torch.bfloat16 else None) if world > 1 else None
    model.engine.grad_sync = sync
    start_epoch, global_step, best_ade = 0, 0, float("inf")
    os.makedirs(args.out_dir, exist_ok=True)
    latest = os.path.join(args.out_dir, "latest_model.pt")
    if getattr(args, "resume", False) and os.path.exists(latest):                      # train.py:138-150
        ck = torch.load(latest, map_location="cpu", weights_only=True)
        model.load_state_dict(ck["model_state_dict"])
        opt.load_state_dict(ck["optimizer_state_dict"])
        start_epoch, global_step, best_ade = ck["epoch"] + 1, ck["global_step"], ck.get("best_ade", best_ade)
        opt.resync_masters()
    steps_per_epoch = len(train_data) // (args.bs * world)
    total_steps = steps_per_epoch * args.epochs
    history = []
    model.train()
    for epoch in range(start_epoch, args.epochs):
        g = np.random.Generator(np.random.Philox(key=np.array([1234, epoch], dtype=np.uint64)))   # same order on every rank
        order = g.permutation(len(train_data))
        run = torch.zeros((), device=device)
        for it in range(steps_per_epoch):
            lo, hi = shard_range(args.bs * world, rank, world)
            idx = order[it * args.bs * world + lo: it * args.bs * world + hi]
            batch = train_data.batch(idx, device, args.max_traj_token)
            loss = model.loss_and_backward(batch["tokens"], batch["attention_masks"], batch["pcrgbs"], batch["prompts"].shape[1],
                                           model.dims.tok.pad, fps_start=torch.zeros(len(idx), dtype=torch.int32, device=device))
            if sync is not None:
                sync.finish()
            opt.step(grad_scale=sync.grad_scale if sync is not None else 1.0, lr=linear_warmup_lr(args.lr_llm, global_step, total_steps))
            run += loss
            global_step += 1
        rec = {"epoch": epoch, "train_loss": float(run) / max(1, steps_per_epoch), "global_step": global_step}
        if val_data is not None:
            m, _ = run_validation(model, val_data, args, device, max_batches=getattr(args, "val_batches", None))
            rec.update(m)
        if world > 1:
            t = torch.tensor([rec["train_loss"]], device=device)
            dist.all_reduce(t)
            rec["train_loss"] = float(t) / world
        history.append(rec)
        if rank == 0:
            log(json.dumps(rec))
            ck = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                  "optimizer_state_dict": opt.state_dict_cpu(), "scheduler_state_dict": {"last_step": global_step}, "global_step": global_step}
            torch.save(ck, latest)                                                      # train.py:287-296
            if val_data is not None and rec.get("ADE", float("inf")) < best_ade:
                best_ade = rec["ADE"]
                ck["best_ade"] = best_ade
                torch.save(ck, os.path.join(args.out_dir, "best_model_ade.pt"))         # train.py:298-308
    return history


def evaluate(args, model, data, split="test", device="cuda"):
    """evaluate.py:70-170: load best_model_ade.pt, generate on the split, dump {split}_gen_trajs.json."""
    best = os.path.join(args.checkpoint_dir, "best_model_ade.pt")
    if os.path.exists(best):
        model.load_state_dict(torch.load(best, map_location="cpu", weights_only=True)["model_state_dict"])
    metrics, dump = run_validation(model, data, args, device)
    rank, _ = _rank_world()
    if rank == 0:
        with open(os.path.join(args.checkpoint_dir, f"{split}_gen_trajs.json"), "w") as f:
            json.dump(dump, f)
    return metrics


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="EgoScaler trajectory generator on MI355X (flags of train.py:312-347)")
    ap.add_argument("mode", choices=["train", "eval"])
    ap.add_argument("--model_name", default=None, help="local HF directory; omit for seeded synthetic weights")
    ap.add_argument("--max_traj_token", type=int, default=160)
    ap.add_argument("--max_desc_token", type=int, default=20)
    ap.add_argument("--num_steps", type=int, default=20)
    ap.add_argument("--do_norm", action="store_true")
    ap.add_argument("--do_standard", action="store_true")
    ap.add_argument("--unfreeze_pc_encoder", action="store_true")
    ap.add_argument("--unfreeze_language_model", action="store_true")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--grad_accum_steps", type=int, default=1)
    ap.add_argument("--lr_llm", type=float, default=2e-5)
    ap.add_argument("--num_bins", type=int, default=256)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    ap.add_argument("--out_dir", default="runs/egoscaler_amd")
    ap.add_argument("--checkpoint_dir", default=None)
    ap.add_argument("--tiny", action="store_true", help="small synthetic model (smoke runs)")
    ap.add_argument("--n_train", type=int, default=64)
    ap.add_argument("--n_val", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    return ap.parse_args(argv)


def main(argv=None):
    from .config import dims_7b, dims_tiny
    from .pointllm import TrajPointLLMForCausalLM
    a = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(a.local_rank)
    dev = torch.device("cuda", a.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    dims = dims_tiny() if a.tiny else dims_7b()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = TrajPointLLMForCausalLM(a, dims, a.model_name, device=dev, dtype=dtype)
    if a.model_name is None:
        sd = synth.synth_state_dict(dims, 0)
        model.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
    a.checkpoint_dir = a.checkpoint_dir or a.out_dir
    val = SyntheticTrajData(dims, a.n_val, num_steps=a.num_steps, seed=977)
    if a.mode == "train":
        train(a, model, SyntheticTrajData(dims, a.n_train, num_steps=a.num_steps), val, dev)
    else:
        print(json.dumps(evaluate(a, model, val, "test", dev)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
