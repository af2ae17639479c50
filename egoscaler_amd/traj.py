"""Trajectory (de)tokenisation, re-sampling, scaling and metrics.

Two layers, mirroring where the reference does this work:
  * batch-level integer work on the device (egomi_traj_tokenize / _detokenize / _metrics):
    what CustomDataset.__getitem__/collate_fn and detokenize_traj do per sample on the host
    (models/pointllm/dataset.py:150-194; the per-sample methods are missing from the release,
    SURVEY.md §0.1, so the contract is built from models/pointllm/utils/utils.py:13-104);
  * small host helpers with the reference's names and semantics for single trajectories / strings
    (utils.py, models/utils/traj_utils.py, models/utils/metrics.py), numpy like the reference.
Nothing here imports oracle/.
"""
import re

import numpy as np
import torch

from ._lib import c_i, c_i64, call
from .ops import P, S

# Aria pin-hole camera and workspace constants (egoscaler/configs/camera.py:7-9, configs/dataset.py:1-7)
PINHOLE_IMAGE_SIZE = 1408
FOCAL_LEN = 605.343
PRINCIPAL_POINT = 703.5
WORKSPACE = {"min_x": -2.0, "max_x": 2.0, "min_y": -2.0, "max_y": 2.0, "min_z": 0.0, "max_z": 2.5}


# ------------------------------------------------------------------------------------------ device
def bin_edges(num_bins: int, device) -> torch.Tensor:
    return torch.from_numpy(np.linspace(-1, 1, num_bins)).to(device)            # float64, numpy's own values


def tokenize_batch(traj: torch.Tensor, tok, seq_len: int, steps=None):
    """traj f32 [B,T,6] in [-1,1] -> (ids i64 [B,seq_len], mask bool [B,seq_len]).
    Layout <ts> (p*6 <tsep>)*T <te> eos pad...; raises ValueError if seq_len is too short."""
    traj = traj.to(torch.float32).contiguous()
    B, T, _ = traj.shape
    dev = traj.device
    ids = torch.empty(B, seq_len, dtype=torch.int64, device=dev)
    mask = torch.empty(B, seq_len, dtype=torch.uint8, device=dev)
    err = torch.empty(B, dtype=torch.int32, device=dev)
    st = None if steps is None else torch.as_tensor(steps, dtype=torch.int32, device=dev).contiguous()
    call("egomi_traj_tokenize", P(traj), P(st), c_i(B), c_i(T), P(bin_edges(tok.num_bins, dev)), c_i(tok.num_bins), c_i64(tok.p0),
         c_i64(tok.ts), c_i64(tok.tsep), c_i64(tok.te), c_i64(tok.eos), c_i64(tok.pad), c_i(seq_len), P(ids), P(mask), P(err), S())
    if int(err.max()) != 0:
        raise ValueError("trajectory tokens exceed max_traj_token")
    return ids, mask.bool()


def detokenize_batch(ids: torch.Tensor, tok, max_steps: int):
    """ids i64 [B,L] (generated span) -> (values f32 [B,max_steps,6] in [-1,1], n_steps i32 [B])."""
    ids = ids.to(torch.int64).contiguous()
    B, L = ids.shape
    dev = ids.device
    out = torch.zeros(B, max_steps, 6, dtype=torch.float32, device=dev)
    n = torch.empty(B, dtype=torch.int32, device=dev)
    call("egomi_traj_detokenize", P(ids), c_i(B), c_i(L), P(bin_edges(tok.num_bins, dev)), c_i(tok.num_bins), c_i64(tok.p0), c_i64(tok.tsep),
         c_i64(tok.eos), c_i(max_steps), P(out), P(n), S())
    return out, n


def metrics_batch(gen, n_gen, gt, n_gt=None):
    """ADE / FDE per sample (float64), generated trajectories padded with their last step or cut to
    the ground-truth length (models/utils/metrics.py:40-52)."""
    gen, gt = gen.to(torch.float32).contiguous(), gt.to(torch.float32).contiguous()
    B, T, D = gt.shape
    if gen.shape != gt.shape:
        raise ValueError("gen and gt must share [B,T,D]")
    dev = gt.device
    ade = torch.empty(B, dtype=torch.float64, device=dev)
    fde = torch.empty(B, dtype=torch.float64, device=dev)
    ng = None if n_gen is None else n_gen.to(torch.int32).contiguous()
    nt = None if n_gt is None else n_gt.to(torch.int32).contiguous()
    call("egomi_traj_metrics", P(gen), P(ng), P(gt), P(nt), c_i(B), c_i(T), c_i(D), P(ade), P(fde), S())
    return ade, fde


# ------------------------------------------------------------------------------------------ host
def discretize_action(action_vector, num_bins=256):
    """utils/utils.py:13-16."""
    return (np.digitize(action_vector, np.linspace(-1, 1, num_bins)) - 1).tolist()


def token_to_action(tokens, num_bins=256):
    """utils/utils.py:18-21."""
    bins = np.linspace(-1, 1, num_bins)
    return [bins[v] for v in tokens]


def rt2_scaler(traj: np.ndarray, maxmin, split=None) -> np.ndarray:
    """utils/utils.py:23-34: rot*pi, z -> [d_min,d_max], xy pixel -> metric through the Aria intrinsics."""
    d_max, d_min = maxmin
    traj[:, [3, 4, 5]] = np.pi * traj[:, [3, 4, 5]]
    traj[:, 2] = (1.0 / 2) * traj[:, 2] + (1.0 / 2)
    traj[:, 2] = (d_max - d_min) * traj[:, 2] + d_min
    for c in (0, 1):
        traj[:, c] = (PINHOLE_IMAGE_SIZE / 2) * traj[:, c] + (PINHOLE_IMAGE_SIZE / 2)
        traj[:, c] = (traj[:, c] - PRINCIPAL_POINT) * traj[:, 2] / FOCAL_LEN
    return traj


def simple_scaler(traj: np.ndarray, maxmin) -> np.ndarray:
    """utils/utils.py:36-45: the <x..><y..><z..><rx..><ry..><rz..> format — rotations in percent of a turn, depth in percent of the
    [d_min, d_max] range, x / y already in pixels."""
    d_max, d_min = maxmin
    traj[:, [3, 4, 5]] = np.pi * (2 * (traj[:, [3, 4, 5]] / 100) - 1)
    traj[:, 2] = traj[:, 2] / 100
    traj[:, 2] = traj[:, 2] * (d_max - d_min) + d_min
    traj[:, 0] = (traj[:, 0] - PRINCIPAL_POINT) * traj[:, 2] / FOCAL_LEN
    traj[:, 1] = (traj[:, 1] - PRINCIPAL_POINT) * traj[:, 2] / FOCAL_LEN
    return traj


_PATTERNS = {                                                                  # utils/utils.py:49-61
    (True, "pos"): re.compile(r"<p(\d+)> <p(\d+)> <p(\d+)>"),
    (True, "xy"): re.compile(r"<p(\d+)> <p(\d+)>"),
    (True, "full"): re.compile(r"<p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)>"),
    (False, "pos"): re.compile(r"<x(\d+)><y(\d+)><z(\d+)>"),
    (False, "full"): re.compile(r"<x(\d+)><y(\d+)><z(\d+)><rx(\d+)><ry(\d+)><rz(\d+)>"),
}


def str_to_float(s, maxmin, split=None, rt2=False, only_pos=False, only_xy=False, z_values=None, num_bins=256):
    """utils/utils.py:47-104, every format: `rt2` (<p*> bins; 6-DoF, `only_pos` = 3 bins with rotations at bin 0, `only_xy` = 2 bins with
    the depth taken from `z_values[i]`, the last one repeated) or the per-axis <x..><y..> form (`only_pos`: 3 integers; `only_xy` is ignored
    there, as in the reference).  One match per `<tsep>` segment, a segment without one repeats the previous step (only once one exists,
    :88-90); float32 array through rt2_scaler / simple_scaler, None when nothing parsed.  (The reference's default is rt2=False; the
    trajectory generator's drivers pass rt2=True.)"""
    kind = "pos" if only_pos else ("xy" if (only_xy and rt2) else "full")
    pattern = _PATTERNS[(bool(rt2), kind)]
    traj, last = [], None
    for i, seg in enumerate(s.split("<tsep>")):
        m = pattern.search(seg)
        if m:
            if rt2:
                g = [int(v) for v in m.groups()]
                x, y, z, rx, ry, rz = token_to_action(g + [0] * (6 - len(g)), num_bins=num_bins)
                if kind == "xy":
                    z = z_values[i] if i < len(z_values) else z_values[-1]
            elif only_pos:
                x, y, z = map(int, m.groups())
                rx, ry, rz = 0, 0, 0
            else:
                x, y, z, rx, ry, rz = map(float, m.groups())
            last = (x, y, z, rx, ry, rz)
            traj.append(last)
        elif last is not None:
            traj.append(last)
    if not traj:
        return None
    traj = np.array(traj).astype(np.float32)
    return rt2_scaler(traj, maxmin, split) if rt2 else simple_scaler(traj, maxmin)


def denorm(traj: np.ndarray) -> np.ndarray:
    """dataset.py:139-145 (do_norm): [-1,1] -> workspace metres, rotations * pi.  traj [B,T,6]."""
    t = np.array(traj, copy=True)
    t[:, :, [0, 1, 2]] = (t[:, :, [0, 1, 2]] + 1) / 2
    for c, k in enumerate("xyz"):
        t[:, :, c] = t[:, :, c] * (WORKSPACE["max_" + k] - WORKSPACE["min_" + k]) + WORKSPACE["min_" + k]
    t[:, :, [3, 4, 5]] *= np.pi
    return t


def preprocess_traj(traj: np.ndarray, num_steps: int, return_padding_mask: bool = False):
    """models/utils/traj_utils.py:3-39."""
    T = traj.shape[0]
    if T >= num_steps:
        out, pm = traj[np.linspace(0, T - 1, num_steps).astype(int)], np.ones(num_steps, dtype=int)
    else:
        out = np.vstack([traj, np.tile(traj[-1], (num_steps - T, 1))])
        pm = np.concatenate([np.ones(T, dtype=int), np.zeros(num_steps - T, dtype=int)])
    return (out, pm) if return_padding_mask else out


def smoothing_traj(traj: np.ndarray) -> np.ndarray:
    """models/utils/traj_utils.py:41-96 (5-tap box filter on xyz with the reference's edge rules)."""
    p, n, out = traj[:, :3], traj.shape[0], []
    for j in range(n):
        if j == 0:
            m = (3 * p[0] + p[1] + p[2]) / 5 if n >= 3 else ((3 * p[0] + p[1]) / 4 if n == 2 else p[0])
        elif j == 1:
            m = (2 * p[0] + p[1] + p[2] + p[3]) / 5 if n >= 4 else ((2 * p[0] + p[1] + p[2]) / 4 if n == 3 else p[1])
        elif j == n - 2:
            m = (p[j - 2] + p[j - 1] + p[j] + p[j + 1]) / 4 if n >= 4 else ((p[j - 1] + p[j] + p[j + 1]) / 3 if n == 3 else p[j])
        elif j == n - 1:
            m = (p[j - 2] + p[j - 1] + p[j]) / 3 if n >= 3 else ((p[j - 1] + p[j]) / 2 if n == 2 else p[j])
        else:
            m = (p[j - 2] + p[j - 1] + p[j] + p[j + 1] + p[j + 2]) / 5
        out.append(m)
    return np.concatenate([np.array(out), traj[:, 3:]], axis=-1)


def _pad_like(gen, gt):
    if gen.shape[0] > gt.shape[0]:
        return gen[:gt.shape[0]]
    if gen.shape[0] < gt.shape[0]:
        return np.vstack([gen, np.repeat(gen[-1].reshape(1, -1), gt.shape[0] - gen.shape[0], axis=0)])
    return gen


def average_displacement_error(gen_traj, gt_traj) -> float:
    """models/utils/metrics.py:38-55 (axis=1 norm, as written; hand it [T,D], or [1,T,D] to get the
    value the reference drivers actually log)."""
    return np.linalg.norm(gt_traj - _pad_like(gen_traj, gt_traj), ord=2, axis=1).mean()


def final_displacement_error(gen_traj, gt_traj) -> float:
    """models/utils/metrics.py:7-27."""
    return np.linalg.norm(gt_traj[-1] - _pad_like(gen_traj, gt_traj)[-1], ord=2)


def initial_displacement_error(gen_traj, gt_traj) -> float:
    """models/utils/metrics.py:29-36."""
    return np.linalg.norm(gt_traj[0] - gen_traj[0], ord=2)


def anglar_distance(gen_rot, gt_rot) -> float:
    """models/utils/metrics.py:61-87 (name as in the reference): mean 2*acos(|<q1,q2>| clipped) over steps."""
    from scipy.spatial.transform import Rotation as R
    g = _pad_like(gen_rot, gt_rot)
    ad = []
    for a, b in zip(g, gt_rot):
        d = np.dot(R.from_rotvec(a).as_quat(), R.from_rotvec(b).as_quat())
        ad.append(2 * np.arccos(np.clip(d, -1.0, 1.0)))
    return sum(ad) / len(ad)


class TargetNorm:
    """Target normalisation of CustomDataset (models/pointllm/dataset.py:41-148): `--do_norm` (workspace bounds,
    rotations / pi) or `--do_standard` (dataset mean / std, then a per-sample max-abs scale so values land in [-1, 1]),
    with the statistics side file `norm_param.json` {"mean", "std"} (dataset.py:104-124).

    `denorm` is the reference's (dataset.py:126-148).  The forward direction lives in the reference's missing
    `__getitem__`; it is restated here as the exact inverse of `denorm`:
      do_norm     : x,y,z -> 2*(v - min)/(max - min) - 1 ; rotvec -> r / pi ;            max_abs = 1
      do_standard : z = (v - mean) / std ; max_abs[d] = max_t |z[t,d]| ; v_norm = z / max_abs
    mode "none" (neither flag; the reference's denorm then returns None): values are taken as already normalised."""

    def __init__(self, do_norm=False, do_standard=False, mean=None, std=None):
        assert not (do_norm and do_standard), "Cannot enable both normalization methods."            # dataset.py:44
        self.mode = "norm" if do_norm else ("standard" if do_standard else "none")
        self.mean = None if mean is None else np.asarray(mean, dtype=np.float64)
        self.std = None if std is None else np.asarray(std, dtype=np.float64)

    # -- statistics (dataset.py:80-124)
    def fit(self, trajs, num_steps, smooth=False):
        """compute_mean_std (dataset.py:80-102): every trajectory resampled to num_steps (optionally smoothed), mean and
        std over (sample, step), std + 1e-8."""
        allt = []
        for t in trajs:
            t = preprocess_traj(np.asarray(t), num_steps)
            if smooth:
                t = smoothing_traj(t)
            allt.append(t)
        allt = np.array(allt)
        self.mean, self.std = allt.mean(axis=(0, 1)), allt.std(axis=(0, 1)) + 1e-8
        return self.mean, self.std

    def save(self, save_dir):
        import json
        import os
        with open(os.path.join(save_dir, "norm_param.json"), "w") as f:
            json.dump({"mean": self.mean.tolist(), "std": self.std.tolist()}, f)

    def load(self, save_dir):
        import json
        import os
        with open(os.path.join(save_dir, "norm_param.json")) as f:
            p = json.load(f)
        self.mean, self.std = np.array(p["mean"]), np.array(p["std"])
        return self

    # -- per sample
    def normalize(self, traj):
        """traj [T,6] (metres, rotation vectors) -> (values in [-1,1] [T,6] float64, max_abs [6])."""
        t = np.array(traj, dtype=np.float64, copy=True)
        if self.mode == "norm":
            for c, k in enumerate("xyz"):
                lo, hi = WORKSPACE["min_" + k], WORKSPACE["max_" + k]
                t[:, c] = 2.0 * (t[:, c] - lo) / (hi - lo) - 1.0
            t[:, 3:6] /= np.pi
            return t, np.ones(6)
        if self.mode == "standard":
            if self.mean is None:
                raise ValueError("do_standard needs statistics: fit() on the train split or load() norm_param.json")
            z = (t - self.mean) / self.std
            m = np.abs(z).max(axis=0)
            m = np.where(m > 0, m, 1.0)
            return z / m, m
        return t, np.ones(6)

    def denorm(self, traj, max_abs):
        """dataset.py:126-148.  traj [B,T,6] (numpy, copied), max_abs [B,6]."""
        t = np.array(traj, copy=True)
        if self.mode == "norm":
            return denorm(t)
        if self.mode == "standard":
            t = t * np.asarray(max_abs)[:, None, :]
            return t * self.std + self.mean
        return t
