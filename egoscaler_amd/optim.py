"""AdamW over the engine's fp32 main_grad buffers (reference optimizer: train.py:107-117,
AdamW lr 2e-5 + linear warm-up over 1/5 of the steps; DeepSpeed keeps fp32 master weights for bf16
params, train.py:97-98 — same here, without ZeRO sharding: 288 GB holds everything replicated)."""
import torch

from . import ops


class EgoAdamW:
    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.t = 0
        self.state = {}
        for n, p in model.named_parameters():
            if p.requires_grad:
                master = p.data if p.dtype == torch.float32 else p.data.float()
                self.state[n] = {"p": p, "master": master, "m": torch.zeros_like(master), "v": torch.zeros_like(master)}

    def arm(self, grad_scale=1.0, lr=None):
        """Call right before the LAST backward pass of an optimizer step (one rank, the step about to follow with step(overlap=True) and the same
        grad_scale / lr): every decoder layer is then updated as soon as ITS gradients are final (Engine._notify_layer), on the side stream, under the
        backward pass of the layers below it — the update is HBM-bound and gets the chip whenever the compute stream runs something that leaves
        registers free (attention, norms, RoPE; the 8-phase GEMM does not), instead of queueing all 38 ms of it behind the backward pass.  step()
        then handles what is left (projector / point backbone, embedding, final norm, lm_head).  Values are the same as the plain step's.
        Returns False (and arms nothing) where that is not possible: an exchange is attached (gradients are not final until it has run), no CUDA
        device, or resident W^T copies of trainable weights that a later product of this backward could still read."""
        eng = self.model.engine
        self._armed = None
        eng.layer_final_hook = None
        if (eng.grad_sync is not None and not getattr(eng.grad_sync, "local", False)) or eng.device.type != "cuda" or not eng.any_layer_trainable:
            return False
        if any(nm in eng.trainable for nm in getattr(eng, "wT", {})):      # (EGOMI_GEMM_TN=0 route; before the first forward pass the hook looks again)
            return False
        groups = {}
        for n in self.state:
            groups.setdefault(eng.param_group_of(n), []).append(n)
        self._armed = {"lr": self.lr if lr is None else lr, "grad_scale": grad_scale, "done": set(), "groups": groups}
        eng.layer_final_hook = self._early_layer
        return True

    def _early_layer(self, l):
        a, eng = self._armed, self.model.engine
        names = a["groups"].get(l)
        if not names or any(nm in eng.trainable for nm in eng.wT):
            return
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())       # the layer's gradients are final; every reader of its old weights has been queued
        with torch.cuda.stream(self._side):
            for n in names:
                self._update(n, self.state[n], eng, a["lr"], a["grad_scale"], t=self.t + 1)
            ev = torch.cuda.Event()
            ev.record(self._side)
            eng.param_events[l] = ev
        a["done"].add(l)

    def _update(self, n, st, eng, lr, grad_scale, t=None):
        g = eng.reduced_grad.get(n)                        # resident exchange (dp.GradSync(resident=True)): the rank-summed bf16 gradient, read
        if g is None:                                      # in place from the layer's wire buffer
            g = eng.main_grad.get(n)
        if g is None:
            return
        p = st["p"]
        copy = None if p.dtype == torch.float32 else p.data
        ops.adamw(st["master"], copy, g, st["m"], st["v"], lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t if t is None else t, grad_scale)

    def step(self, grad_scale=1.0, lr=None, overlap=False):
        """overlap=True (training loops that go straight on to the next forward pass: bench.py, driver.train): the updates run on a side stream in
        the order the NEXT forward pass needs the tensors — projector / point backbone, embedding, decoder layers 0..L-1, final norm + lm_head — with
        one event per group, and the engine's forward waits for a group right before it first reads it (Engine.wait_params).  The pass is HBM-bound
        (30 B per parameter) and shares the CUs with the MFMA-bound forward products instead of running alone between two steps; values are the
        same.  Whoever reads parameters or optimizer state outside a forward pass calls engine.wait_param_updates() first (state_dict, generate,
        checkpoints and this class's own state accessors do)."""
        self.t += 1
        lr = self.lr if lr is None else lr
        eng = self.model.engine
        armed, self._armed = getattr(self, "_armed", None), None
        eng.layer_final_hook = None
        early = armed["done"] if armed is not None else set()       # decoder layers arm() has already updated under the backward pass
        if early and (armed["lr"] != lr or armed["grad_scale"] != grad_scale):
            raise RuntimeError("EgoAdamW.step: lr / grad_scale differ from the ones given to arm()")
        if not (overlap and eng.device.type == "cuda" and not any(nm in eng.trainable for nm in eng.wT)):
            eng.wait_param_updates()
            for n, st in self.state.items():
                if eng.param_group_of(n) not in early:
                    self._update(n, st, eng, lr, grad_scale)
            eng.after_weights_update()
            return
        groups = {}
        for n in self.state:
            groups.setdefault(eng.param_group_of(n), []).append(n)
        L = eng.dims.lm.num_hidden_layers
        order = ["pre", "embed"] + list(range(L)) + ["post"]
        cur = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(cur)                        # gradients final (and exchanged); every reader of the old weights has been queued
        with torch.cuda.stream(self._side):
            for key in order:
                names = groups.get(key)
                if not names or key in early:
                    continue
                for n in names:
                    self._update(n, self.state[n], eng, lr, grad_scale)
                ev = torch.cuda.Event()
                ev.record(self._side)
                eng.param_events[key] = ev
        eng.after_weights_update()

    def zero_grad(self):
        self.model.engine.zero_grad()
        for st in self.state.values():
            st["p"].grad = None

    def state_dict(self):
        self.model.engine.wait_param_updates()
        return {"t": self.t, "state": {n: {k: st[k] for k in ("master", "m", "v")} for n, st in self.state.items()}}

    def state_dict_cpu(self):
        self.model.engine.wait_param_updates()
        return {"t": self.t, "state": {n: {k: st[k].detach().cpu() for k in ("master", "m", "v")} for n, st in self.state.items()}}

    def resync_masters(self):
        """After loading a checkpoint: the low-precision model copies follow the fp32 masters again."""
        self.model.engine.wait_param_updates()
        for st in self.state.values():
            if st["p"].dtype != torch.float32:
                st["p"].data.copy_(st["master"])
        self.model.engine.after_weights_update()

    def load_state_dict(self, sd):
        self.model.engine.wait_param_updates()
        self.t = sd["t"]
        for n, s in sd["state"].items():
            for k in ("master", "m", "v"):
                self.state[n][k].copy_(s[k])


def linear_warmup_lr(base_lr, step, total_steps, warmup_steps=None):
    """HF `get_linear_schedule_with_warmup` as the reference sets it up (train.py:113-116): warm-up over
    int(total/5) steps starting from lr = 0 at step 0 (`step / warm`), then a linear decay that reaches 0 at
    `total_steps`.  `step` is the number of optimizer steps already taken (LambdaLR's epoch counter)."""
    warm = int(total_steps / 5) if warmup_steps is None else int(warmup_steps)
    if step < warm:
        return base_lr * float(step) / float(max(1, warm))
    return base_lr * max(0.0, float(total_steps - step) / float(max(1, total_steps - warm)))
