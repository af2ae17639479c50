"""Oracle for SURVEY.md §8a row A14: trajectory (de)tokenisation, re-sampling and metrics.
Test infrastructure only (see oracle/__init__.py).

Status per function:
  * preprocess_traj / smoothing_traj / ADE / FDE: the reference modules
    egoscaler/models/utils/{traj_utils,metrics}.py import here (metrics with the fastdtw stand-in);
    oracle/gen_golden.py records their outputs -> PINNED.
  * discretize / token_to_action / rt2_scaler / parse_traj_string / denorm: the reference module
    egoscaler/models/pointllm/utils/utils.py raises AttributeError at import (camera_cfg has
    `focal_len`, the module reads `focal_length`, utils.py:10) and dataset.py needs missing deps, so
    these are RESTATED FROM TEXT and pinned by numpy known-answers only ("restated-from-text").
"""
import re

import numpy as np

ARIA_SIZE = 1408          # egoscaler/configs/camera.py:7
ARIA_FOCAL = 605.343      # camera.py:8 (named focal_len there)
ARIA_PP = 703.5           # camera.py:9
WORKSPACE = dict(min_x=-2.0, max_x=2.0, min_y=-2.0, max_y=2.0, min_z=0.0, max_z=2.5)   # configs/dataset.py:1-7


def discretize_action(v, num_bins=256):
    """utils/utils.py:13-16. values < -1 -> -1, >= 1 -> num_bins-1."""
    return (np.digitize(v, np.linspace(-1, 1, num_bins)) - 1).tolist()


def token_to_action(tokens, num_bins=256):
    """utils/utils.py:18-21."""
    bins = np.linspace(-1, 1, num_bins)
    return [bins[t] for t in tokens]


def rt2_scaler(traj: np.ndarray, maxmin) -> np.ndarray:
    """utils/utils.py:23-34 (in place on a float32 array, as the reference does)."""
    d_max, d_min = maxmin
    traj[:, [3, 4, 5]] = np.pi * traj[:, [3, 4, 5]]
    traj[:, 2] = 0.5 * traj[:, 2] + 0.5
    traj[:, 2] = (d_max - d_min) * traj[:, 2] + d_min
    traj[:, 0] = (ARIA_SIZE / 2) * traj[:, 0] + (ARIA_SIZE / 2)
    traj[:, 0] = (traj[:, 0] - ARIA_PP) * traj[:, 2] / ARIA_FOCAL
    traj[:, 1] = (ARIA_SIZE / 2) * traj[:, 1] + (ARIA_SIZE / 2)
    traj[:, 1] = (traj[:, 1] - ARIA_PP) * traj[:, 2] / ARIA_FOCAL
    return traj


_PAT6 = re.compile(r"<p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)> <p(\d+)>")


def parse_traj_string(s: str, num_bins=256):
    """utils/utils.py:47-104 with rt2=True, full 6-DoF: split on <tsep>, first regex match per
    segment, unmatched segments repeat the previous step (only once one exists, :88-90).
    Returns float32 [T,6] in [-1,1] bin values, or None."""
    traj, last = [], None
    for seg in s.split("<tsep>"):
        m = _PAT6.search(seg)
        if m:
            cur = tuple(token_to_action([int(g) for g in m.groups()], num_bins))
            traj.append(cur)
            last = cur
        elif last is not None:
            traj.append(last)
    if not traj:
        return None
    return np.array(traj).astype(np.float32)


def denorm_workspace(traj: np.ndarray) -> np.ndarray:
    """dataset.py:139-145 (do_norm branch): [-1,1] -> metric workspace, rotations * pi. [B,T,6]."""
    t = traj.copy()
    t[:, :, [0, 1, 2]] = (t[:, :, [0, 1, 2]] + 1) / 2
    t[:, :, 0] = t[:, :, 0] * (WORKSPACE["max_x"] - WORKSPACE["min_x"]) + WORKSPACE["min_x"]
    t[:, :, 1] = t[:, :, 1] * (WORKSPACE["max_y"] - WORKSPACE["min_y"]) + WORKSPACE["min_y"]
    t[:, :, 2] = t[:, :, 2] * (WORKSPACE["max_z"] - WORKSPACE["min_z"]) + WORKSPACE["min_z"]
    t[:, :, [3, 4, 5]] *= np.pi
    return t


def preprocess_traj(traj: np.ndarray, num_steps: int):
    """models/utils/traj_utils.py:3-39."""
    T = traj.shape[0]
    if T >= num_steps:
        return traj[np.linspace(0, T - 1, num_steps).astype(int)]
    return np.vstack([traj, np.tile(traj[-1], (num_steps - T, 1))])


def smoothing_traj(traj: np.ndarray) -> np.ndarray:
    """models/utils/traj_utils.py:41-96: asymmetric 5-tap box filter on xyz with the edge rules."""
    p = traj[:, :3]
    n = p.shape[0]
    out = []
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


def ade(gen: np.ndarray, gt: np.ndarray) -> float:
    """models/utils/metrics.py:38-55, on [T,D] inputs (the documented form)."""
    return float(np.linalg.norm(gt - _pad_like(gen, gt), ord=2, axis=1).mean())


def fde(gen: np.ndarray, gt: np.ndarray) -> float:
    """models/utils/metrics.py:7-27."""
    return float(np.linalg.norm(gt[-1] - _pad_like(gen, gt)[-1], ord=2))


def ade_as_called(gen: np.ndarray, gt: np.ndarray) -> float:
    """What train.py:258 / evaluate.py:144 actually compute: the function is handed [1,T,6], so
    axis=1 is TIME: mean over the 6 dims of the L2 norm over time."""
    return float(np.linalg.norm(gt[None] - gen[None], ord=2, axis=1).mean())
