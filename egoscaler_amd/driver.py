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
the device (`build_batch`).  Data sources of the command line: EgoScaler files (`--root_dir/--data_dir`: data_io.EgoScalerFiles +
FileTrajData, descriptions through the HF tokenizer of `--model_name` with the reference's prompt template, dataset.py:16-19) or, without
them, the seeded synthetic samples of SURVEY.md §8d.  `--model_name DIR` goes through `build_model(args)` exactly as train.py:67 /
evaluate.py:79 do (config.json, weights and tokenizer of the directory; trajectory tokens appended, builder.py:33-46).
"""
import argparse
import json
import os
import types

import numpy as np
import torch
import torch.distributed as dist

from . import ops, synth, traj as T
from .dp import GradSync, shard_range, ragged_shard_range
from .optim import EgoAdamW, linear_warmup_lr


def collate(dims, image_ids, pcrgbs, desc_tokens, desc_masks, traj_tokens, traj_masks, gt_trajs, max_abs, sep_ids=()):
    """`CustomDataset.collate_fn` (dataset.py:150-194) on device tensors that are already stacked:
    tokens = desc | <sep> ids | traj ; masks likewise (the <sep> ids are always attended) ; prompts run up to and including
    the first <tsep> of SAMPLE 0 (dataset.py:180-182) ; 'trajectory_masks' is the token-level trajectory mask."""
    dev = pcrgbs.device
    B = desc_tokens.shape[0]
    sep = torch.as_tensor(list(sep_ids), dtype=torch.int64, device=dev).reshape(1, -1).expand(B, -1)
    tokens = torch.cat((desc_tokens, sep, traj_tokens), dim=-1)
    masks = torch.cat((desc_masks.bool(), torch.ones_like(sep), traj_masks.bool()), dim=-1)      # bool ++ int64 -> int64 0/1, as in the reference
    hit = (tokens[0] == dims.tok.tsep).nonzero()
    if hit.numel() == 0:
        raise IndexError("no <tsep> in the first sample")                  # the reference's [1][0] raises IndexError here too
    pos = int(hit[0, 0])
    return {"image_ids": image_ids, "pcrgbs": pcrgbs, "prompts": tokens[:, :pos + 1], "prompt_masks": masks[:, :pos + 1],
            "tokens": tokens, "attention_masks": masks, "trajectories": gt_trajs, "trajectory_masks": traj_masks.bool(),
            "max_abs": max_abs}


def build_batch(dims, desc_ids, trajs, pcrgbs, max_traj_token=160, image_ids=None, steps=None, desc_mask=None, max_abs=None,
                sep_ids=(), gt_trajs=None):
    """The per-sample half the release lacks (`CustomDataset.__getitem__`) + collate, on the device.
    desc_ids i64 [B,Ld] (description tokens, padded), desc_mask bool [B,Ld] (False on padding; None = all real),
    trajs f32 [B,T,6] in [-1,1] (already normalised), pcrgbs f32 [B,N,6].
    Description part: [bos] desc_a <point_start> <point_patch>xP <point_end> desc_b; trajectory part:
    <ts> (p*6 <tsep>)*T <te> eos pad...  Padding positions of the description carry mask False into `attention_masks`
    and `prompt_masks` exactly like the tokenizer's `desc_masks` do in the reference (dataset.py:161-177)."""
    tok, P = dims.tok, dims.pb.point_token_len
    dev = pcrgbs.device
    B, Ld = desc_ids.shape
    dm = torch.ones(B, Ld, dtype=torch.bool, device=dev) if desc_mask is None else desc_mask.to(dev).bool()
    h = Ld // 2
    col = lambda v: torch.full((B, 1), v, dtype=torch.int64, device=dev)
    one = torch.ones(B, 1, dtype=torch.bool, device=dev)
    head = torch.cat([col(tok.bos), desc_ids[:, :h], col(tok.point_start), torch.full((B, P), tok.point_patch, dtype=torch.int64, device=dev),
                      col(tok.point_end), desc_ids[:, h:]], 1)
    head_mask = torch.cat([one, dm[:, :h], one, torch.ones(B, P, dtype=torch.bool, device=dev), one, dm[:, h:]], 1)
    tt, tm = T.tokenize_batch(trajs, tok, max_traj_token, steps=steps)
    return collate(dims, image_ids if image_ids is not None else torch.arange(B, device=dev), pcrgbs, head, head_mask, tt, tm,
                   trajs if gt_trajs is None else gt_trajs, torch.ones(B, 6, device=dev) if max_abs is None else max_abs, sep_ids)


class SyntheticTrajData:
    """Seeded synthetic samples: RGB-D clip -> un-projection -> 8192-point cloud (A1, A2 on the device), uniform description
    ids of varying length (padded to `max_desc_token` with mask False when that is set), smooth random trajectories in
    workspace metres / rotation vectors that go through `norm` (traj.TargetNorm: --do_norm / --do_standard / neither)."""

    def __init__(self, dims, n_samples, frames=2, size=64, text_len=16, num_steps=20, seed=42, norm=None, max_desc_token=None,
                 ragged_text=False):
        self.dims, self.n, self.frames, self.size, self.text_len, self.num_steps, self.seed = dims, n_samples, frames, size, text_len, num_steps, seed
        self.norm = norm if norm is not None else T.TargetNorm()
        self.max_desc, self.ragged = max_desc_token, ragged_text

    def __len__(self):
        return self.n

    def raw_traj(self, i):
        """[num_steps, 6]: in [-1,1] for mode 'none', else workspace metres + rotation vectors (radians)."""
        g = synth._rng(self.seed + int(i), 0xDA7A)
        desc = g.integers(3, min(self.dims.tok.point_patch, self.dims.lm.vocab_size), size=self.text_len)
        base = g.uniform(-0.6, 0.6, size=(1, 6))
        walk = np.clip(base + np.cumsum(g.normal(0, 0.03, size=(self.num_steps, 6)), 0), -1, 1)
        n_real = int(g.integers(max(2, self.text_len // 2), self.text_len + 1)) if self.ragged else self.text_len
        return desc, n_real, (walk if self.norm.mode == "none" else T.denorm(walk[None])[0])

    def fit_norm(self, save_dir=None):
        """train split with --do_standard: compute_mean_std + norm_param.json (dataset.py:58-66,80-111)."""
        self.norm.fit([self.raw_traj(i)[2] for i in range(self.n)], self.num_steps)
        if save_dir is not None:
            self.norm.save(save_dir)
        return self.norm

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
        Ld = self.text_len if self.max_desc is None else int(self.max_desc)
        Ld += Ld % 2
        desc = np.full((len(idx), Ld), dims.tok.pad, dtype=np.int64)
        dmask = np.zeros((len(idx), Ld), dtype=bool)
        trs, gts, mabs = [], [], []
        for j, i in enumerate(idx):
            d, n_real, raw = self.raw_traj(i)
            n_real = min(n_real, Ld)                                   # --max_desc_token truncates (dataset.py:48)
            desc[j, :n_real], dmask[j, :n_real] = d[:n_real], True
            v, m = self.norm.normalize(raw)
            trs.append(np.clip(v, -1, 1))
            gts.append(raw)
            mabs.append(m)
        to = lambda a, dt: torch.from_numpy(np.stack(a).astype(dt)).to(device)
        return build_batch(dims, torch.from_numpy(desc).to(device), to(trs, np.float32), pc, max_traj_token,
                           image_ids=torch.as_tensor([int(i) for i in idx], device=device), desc_mask=torch.from_numpy(dmask).to(device),
                           max_abs=to(mabs, np.float32), gt_trajs=to(gts, np.float32))


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def micro_batch_per_rank(bs, grad_accum_steps, world):
    """DeepSpeed arithmetic of train.py:92-96: train_batch_size = bs = micro * grad_accum_steps * world."""
    return max(1, -(-int(bs) // (max(1, int(grad_accum_steps)) * max(1, int(world)))))


@torch.no_grad()
def run_validation(model, data, args, device, max_batches=None):
    """Generation + metrics on the rank's shard (train.py:207-264, evaluate.py:104-154): prompts up to the first <tsep>,
    `generate` of the remaining positions (sampling with the reference's defaults top_k 50 / top_p 0.95 / T 1.0 unless
    args.val_sample is False -> greedy), cut at eos, de-tokenise, de-normalise (dataset.denorm), pad with the last step,
    ADE / FDE / GD; sums are all-reduced over ranks so that every rank (and the best-checkpoint choice) sees the same means.

    Metrics: 'ADE'/'FDE' use the documented [T,D] form of metrics.py:38-55,7-27; 'ADE_as_called' is what the reference's
    drivers log (they pass [1,T,6], so the norm runs over TIME: SURVEY.md §0.1); 'GD' is metrics.py:61-87 on the rotation
    vectors [T,3] (the reference's own call passes [1,T,6] and raises inside scipy)."""
    dims = model.dims
    rank, world = _rank_world()
    norm = getattr(data, "norm", None) or T.TargetNorm()
    sample = bool(getattr(args, "val_sample", True))
    per = micro_batch_per_rank(args.bs, 1, world)
    model.eval()
    sums = np.zeros(5)                                   # ADE, FDE, ADE_as_called, GD, n
    dump = {}
    # the reference's val / test DataLoader (train.py:79-82, evaluate.py:93-100: batch_size=bs, shuffle=False, NO drop_last) keeps the
    # short last batch: every sample is generated for and lands in the dump.  A batch is dealt over the ranks; the short one is split
    # raggedly and a rank whose share is empty skips it (the only collective of this pass is the all-reduce of the sums below)
    gb = per * world
    nb = -(-len(data) // gb)
    for bi in range(nb if max_batches is None else min(nb, max_batches)):
        n_here = min(gb, len(data) - bi * gb)
        lo, hi = shard_range(gb, rank, world) if n_here == gb else ragged_shard_range(n_here, rank, world)
        idx = list(range(bi * gb + lo, bi * gb + hi))
        if not idx:
            continue
        batch = data.batch(idx, device, args.max_traj_token)
        prompts, tokens = batch["prompts"], batch["tokens"]
        max_new = tokens.shape[1] - prompts.shape[1]
        out = model.generate(input_ids=prompts, attention_mask=batch["prompt_masks"], point_clouds=batch["pcrgbs"], max_length=max_new,
                             do_sample=sample, fps_start=torch.zeros(len(idx), dtype=torch.int32, device=device))
        gen_ids = out.sequences[:, prompts.shape[1]:]
        # the prompt holds the first step; prepend its six tokens + <tsep> so step 0 is parsed like the rest
        vals, n = T.detokenize_batch(torch.cat([prompts[:, -7:], gen_ids], 1), dims.tok, args.num_steps + 4)
        gt = batch["trajectories"]
        Tn = gt.shape[1]
        n = torch.clamp(n, max=Tn)
        vals_h, n_h = vals[:, :Tn].cpu().numpy(), n.cpu().numpy()
        for j in range(len(idx)):                                           # pad with the last parsed step (train.py:252-256)
            if 0 < n_h[j] < Tn:
                vals_h[j, n_h[j]:] = vals_h[j, n_h[j] - 1]
        gen = norm.denorm(vals_h, batch["max_abs"].cpu().numpy())
        gen_d = torch.from_numpy(np.ascontiguousarray(gen, dtype=np.float32)).to(device)
        ade, fde = T.metrics_batch(gen_d, None, gt)
        ade, fde, gt_h = ade.cpu().numpy(), fde.cpu().numpy(), gt.cpu().numpy()
        ids_h = batch["image_ids"].cpu().numpy()
        for j in range(len(idx)):
            if n_h[j] <= 0:
                continue                                                     # detokenize_traj returned None (train.py:249-250)
            sums += [ade[j], fde[j], T.average_displacement_error(gen[j][None], gt_h[j][None]),
                     T.anglar_distance(gen[j][:, 3:6].astype(np.float64), gt_h[j][:, 3:6].astype(np.float64)), 1.0]
            dump[int(ids_h[j])] = gen[j].tolist()                            # evaluate.py:150: keyed by image_id.item(), not by the dataset index
    if world > 1:
        t = torch.from_numpy(sums).to(device)
        dist.all_reduce(t)
        sums = t.cpu().numpy()
    model.train()
    k = max(sums[4], 1.0)
    nan = float("nan")
    mean = lambda i: float(sums[i] / k) if sums[4] else nan            # plain floats: the records go into checkpoints read with weights_only=True
    return {"ADE": mean(0), "FDE": mean(1), "ADE_as_called": mean(2), "GD": mean(3), "n": int(sums[4])}, dump


def train(args, model, train_data, val_data=None, device="cuda", log=print, step_log=None):
    """train.py:129-310.  Returns the list of per-epoch records.  step_log (optional): called once per optimizer step with
    {"epoch", "step", "learning_rate", "loss"} — the reference's per-step wandb.log (train.py:186-193; reading the loss syncs, as its
    `loss.item()` does).

    Batch arithmetic (train.py:92-96): `--bs` is the optimizer batch of the whole job; each rank runs
    `--grad_accum_steps` micro-batches of ceil(bs / accum / world) samples per optimizer step, gradients accumulate in the
    fp32 main_grad buffers, ONE gradient sync + ONE AdamW step follow the last micro-batch (grad scale 1/(accum*world)).
    LR: HF linear schedule with warm-up over int(total/5) steps then decay to 0 (train.py:113-116), advanced once per
    optimizer step, `total` = optimizer steps of the run.  DEVIATION for --grad_accum_steps > 1: the reference sizes the
    schedule in LOADER iterations (num_training_steps = epochs * len(train_dataloader), train.py:114-116) while DeepSpeed advances it
    once per accumulation boundary, so there the warm-up lasts accum x longer and the rate never decays to zero (it ends at
    1 - 1/accum of the way down).  Here the schedule always spans the optimizer steps actually taken; with accum == 1 (the
    reference's default) the two agree step by step (tests/test_host_glue.py pins both statements against HF's scheduler)."""
    rank, world = _rank_world()
    accum = max(1, int(getattr(args, "grad_accum_steps", 1) or 1))
    micro = micro_batch_per_rank(args.bs, accum, world)
    opt = EgoAdamW(model, lr=float(args.lr_llm))
    sync = GradSync(wire_dtype=torch.bfloat16 if model.engine.dtype == torch.bfloat16 else None, resident=True) if world > 1 else None     # EgoAdamW reads the wire
    if sync is None and model.engine.dtype == torch.bfloat16 and model.engine.any_layer_trainable and model.engine.device.type == "cuda":
        sync = GradSync(wire_dtype=torch.bfloat16, resident=True, local=True)      # one rank, --unfreeze_language_model: bf16 weight gradients in per-layer wire buffers (dp.py), nothing exchanged
    start_epoch, global_step, best_ade = 0, 0, float("inf")
    os.makedirs(args.out_dir, exist_ok=True)
    latest = os.path.join(args.out_dir, "latest_model.pt")
    if getattr(args, "resume", False) and os.path.exists(latest):                      # train.py:138-150
        ck = torch.load(latest, map_location="cpu", weights_only=True)
        model.load_state_dict(ck["model_state_dict"])
        opt.load_state_dict(ck["optimizer_state_dict"])
        start_epoch, global_step, best_ade = ck["epoch"] + 1, ck["global_step"], ck.get("best_ade", best_ade)
        opt.resync_masters()
    if getattr(train_data, "norm", None) is not None and train_data.norm.mode == "standard":     # dataset.py:58-66
        if train_data.norm.mean is None:
            train_data.fit_norm(args.out_dir if rank == 0 else None)
        if val_data is not None and getattr(val_data, "norm", None) is not train_data.norm:
            val_data.norm = train_data.norm
    per_step = micro * accum * world                                                   # samples per optimizer step (== bs when divisible)
    # the reference's train DataLoader (train.py:72-77: batch_size=bs, shuffle=True, NO drop_last) keeps the short last batch of an epoch:
    # len(train_dataloader) = ceil(n / bs) iterations, which is also what sizes its LR schedule (train.py:114-116)
    steps_per_epoch = -(-len(train_data) // per_step)
    total_steps = steps_per_epoch * args.epochs
    history = []
    model.train()
    for epoch in range(start_epoch, args.epochs):
        g = np.random.Generator(np.random.Philox(key=np.array([1234, epoch], dtype=np.uint64)))   # same order on every rank
        order = g.permutation(len(train_data))
        run = torch.zeros((), device=device)
        for it in range(steps_per_epoch):
            # the epoch's last optimizer step may be short (no drop_last, see above).  Its samples are dealt over the micro-batches in the
            # usual order; a micro-batch that does not divide over the ranks is filled up by wrapping around to the start of the epoch's
            # permutation, which is what torch's DistributedSampler(drop_last=False) does, so every rank always runs the SAME number of
            # micro-batches with at least one sample each and joins the same collectives (with one rank nothing is ever padded: the short
            # batch is exactly the reference's)
            n_step = min(per_step, len(train_data) - it * per_step)
            n_mb = min(accum, -(-n_step // (micro * world)))
            for a in range(n_mb):
                base = it * per_step + a * micro * world
                n_micro = min(micro * world, n_step - a * micro * world)
                mb = order[base: base + n_micro]
                if n_micro % world:
                    mb = np.concatenate([mb, order[: world - n_micro % world]])
                lo, hi = shard_range(len(mb), rank, world)
                idx = mb[lo:hi]
                batch = train_data.batch(idx, device, args.max_traj_token)
                last = a == n_mb - 1
                model.accumulate_grads = a > 0                                      # first micro-batch overwrites (optimizer.zero_grad(), train.py:159)
                model.engine.grad_sync = sync if last else None                     # reduce once, after the last micro-batch
                if last and (sync is None or sync.local):                           # one rank: trainable decoder layers are updated under this backward pass
                    opt.arm(grad_scale=1.0 / (n_mb * world), lr=linear_warmup_lr(float(args.lr_llm), global_step, total_steps))
                loss = model.loss_and_backward(batch["tokens"], batch["attention_masks"], batch["pcrgbs"], batch["prompts"].shape[1],
                                               model.dims.tok.pad, fps_start=torch.zeros(len(idx), dtype=torch.int32, device=device))
                run += loss / n_mb
            model.accumulate_grads = False
            if sync is not None:
                sync.finish()
            lr_now = linear_warmup_lr(float(args.lr_llm), global_step, total_steps)
            opt.step(grad_scale=1.0 / (n_mb * world), lr=lr_now, overlap=model.engine.any_layer_trainable)    # --unfreeze_language_model: on a side stream,
                                                                                                              # under the next step's forward pass (optim.py)
            if step_log is not None:
                step_log({"epoch": epoch, "step": global_step, "learning_rate": lr_now, "loss": float(loss) if n_mb == 1 else None})
            global_step += 1
        rec = {"epoch": epoch, "train_loss": float(run) / max(1, steps_per_epoch), "global_step": global_step,
               "learning_rate": linear_warmup_lr(float(args.lr_llm), global_step, total_steps)}
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
            if val_data is not None and rec.get("ADE", float("inf")) < best_ade:        # metrics are all-reduced: same choice on any rank
                best_ade = rec["ADE"]
                ck["best_ade"] = best_ade
                torch.save(ck, os.path.join(args.out_dir, "best_model_ade.pt"))         # train.py:298-308
    model.engine.grad_sync = None
    return history


def evaluate(args, model, data, split="test", device="cuda"):
    """evaluate.py:70-170: load best_model_ade.pt, generate on the split, dump {split}_gen_trajs.json
    ({image_id: [[x,y,z,rx,ry,rz], ...]}, evaluate.py:150,167-170); --do_standard statistics come from
    `{checkpoint_dir}/norm_param.json` (evaluate.py:91 hands checkpoint_dir to the dataset as save_dir)."""
    best = os.path.join(args.checkpoint_dir, "best_model_ade.pt")
    if os.path.exists(best):
        model.load_state_dict(torch.load(best, map_location="cpu", weights_only=True)["model_state_dict"])
    norm = getattr(data, "norm", None)
    if norm is not None and norm.mode == "standard" and norm.mean is None:
        norm.load(args.checkpoint_dir)
    metrics, dump = run_validation(model, data, args, device)
    rank, world = _rank_world()
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, dump)
        dump = {k: v for p in parts for k, v in p.items()}
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
    ap.add_argument("--val_greedy", dest="val_sample", action="store_false",
                    help="validate with greedy decoding instead of the reference's sampling defaults (model_arch.py:82-88)")
    ap.add_argument("--root_dir", default=None, help="EgoScaler data root (pcrgbs/, trajs/ ...: dataset.py:36, dataset_base.py:68-103)")
    ap.add_argument("--data_dir", default=None, help="directory of the split files {train,val,test}.json (dataset.py:37)")
    ap.add_argument("--smooth_traj", action="store_true", help="smoothing_traj on the resampled tracks (dataset.py:39 reads this attribute)")
    ap.add_argument("--split", default="test", help="eval mode: the split to generate for (evaluate.py:89-100 uses 'test')")
    return ap.parse_args(argv)


DESC2TRAJ = {"desc": "Action description: {desc}"}                      # dataset.py:16-19


def make_data(a, dims, tokenizer, split, norm, n_synth, seed):
    """The split's data source: files when --root_dir/--data_dir are given (they need the tokenizer of --model_name), else synthetic."""
    if a.root_dir and a.data_dir:
        if tokenizer is None:
            raise ValueError("--root_dir/--data_dir need --model_name DIR (descriptions are tokenised with its tokenizer)")
        from .data_io import EgoScalerFiles, FileTrajData
        from .pointllm.constant import SEP_TOKEN
        encode = lambda text: tokenizer(DESC2TRAJ["desc"].format(desc=text), add_special_tokens=False).input_ids
        sep = tokenizer(SEP_TOKEN, add_special_tokens=False).input_ids              # dataset.py:54
        return FileTrajData(dims, EgoScalerFiles(a.root_dir, a.data_dir, split), encode, num_steps=a.num_steps, max_desc_token=a.max_desc_token,
                            smooth=a.smooth_traj, norm=norm, sep_ids=tuple(int(t) for t in sep))
    return SyntheticTrajData(dims, n_synth, num_steps=a.num_steps, seed=seed, norm=norm, max_desc_token=a.max_desc_token, ragged_text=True)


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
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    tokenizer = None
    if a.model_name is not None:                                        # train.py:67 / evaluate.py:79: everything comes from the directory
        from .pointllm import build_model
        margs = types.SimpleNamespace(**{**vars(a), "dtype": dtype, "device": dev})
        model, tokenizer, _, _ = build_model(margs)
        dims = model.dims
    else:
        dims = dims_tiny() if a.tiny else dims_7b()
        model = TrajPointLLMForCausalLM(a, dims, None, device=dev, dtype=dtype)
        sd = synth.synth_state_dict(dims, 0)
        model.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
    a.checkpoint_dir = a.checkpoint_dir or a.out_dir
    norm = T.TargetNorm(a.do_norm, a.do_standard)
    if a.mode == "train":
        val = make_data(a, dims, tokenizer, "val", norm, a.n_val, 977)
        train(a, model, make_data(a, dims, tokenizer, "train", norm, a.n_train, 42), val, dev)
    else:
        print(json.dumps(evaluate(a, model, make_data(a, dims, tokenizer, a.split, norm, a.n_val, 977), a.split, dev)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
