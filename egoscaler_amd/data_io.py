"""On-disk formats either side of the path.

Reader / writer for the layout the reference's data pipeline produces and its loaders consume:
  * `{data_dir}/{split}.json`  COCO-like {"images": [...], "annotations": [...]}
        (egoscaler/data/README.md:10-36; models/utils/dataset_base.py:31-39)
  * `{root}/trajs/{take}/{file}.pkl|.pickle`  {"init_bbox", "traj_quat" [n,7], "traj_rotvec" [n,6]}
        (data/train/7_get_object_trajectory.py:324-328; read at dataset_base.py:97-102)
  * `{root}/pcrgbs/{take}/{file}.npy` [N,6], `{root}/depths/...npy`, `{root}/obs_images/...jpg`
  * `{save_dir}/norm_param.json` {"mean", "std"}  (models/pointllm/dataset.py:104-124)
Trajectory pickles are read with a RESTRICTED unpickler: only numpy array reconstruction and builtin
containers are allowed, nothing from the file can execute.
"""
import io
import json
import os
import pickle

import numpy as np

from . import traj as T

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("collections", "OrderedDict"),
}


class _NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name}: trajectory files may only contain numpy arrays")


def load_traj_file(path: str) -> dict:
    """{init_bbox, traj_quat, traj_rotvec} from .pkl/.pickle (restricted) or .npz."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            obj = {k: z[k] for k in z.files}
    else:
        with open(path, "rb") as f:
            obj = _NumpyOnlyUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(obj, dict) or "traj_rotvec" not in obj:
        raise ValueError(f"{path}: not an EgoScaler trajectory file")
    if "traj_quat" not in obj and "traj" in obj:                  # assets/demo/trajectory.pkl names the quaternion track `traj`
        obj["traj_quat"] = obj["traj"]
    return obj


def save_traj_file(path: str, init_bbox, traj_quat, traj_rotvec):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = {"init_bbox": np.asarray(init_bbox), "traj_quat": np.asarray(traj_quat), "traj_rotvec": np.asarray(traj_rotvec)}
    if path.endswith(".npz"):
        np.savez(path, **d)
    else:
        with open(path, "wb") as f:
            pickle.dump(d, f)


def save_norm_params(save_dir, mean, std):
    with open(os.path.join(save_dir, "norm_param.json"), "w") as f:
        json.dump({"mean": np.asarray(mean).tolist(), "std": np.asarray(std).tolist()}, f)


def load_norm_params(save_dir):
    with open(os.path.join(save_dir, "norm_param.json")) as f:
        p = json.load(f)
    return np.array(p["mean"]), np.array(p["std"])


def normalize_workspace(traj: np.ndarray) -> np.ndarray:
    """Inverse of dataset.py:139-145 (`do_norm`): metres / radians -> [-1, 1]."""
    t = np.array(traj, dtype=np.float64, copy=True)
    for c, k in enumerate("xyz"):
        lo, hi = T.WORKSPACE["min_" + k], T.WORKSPACE["max_" + k]
        t[..., c] = 2.0 * (t[..., c] - lo) / (hi - lo) - 1.0
    t[..., 3:6] /= np.pi
    return t


class EgoScalerFiles:
    """Index of one split (dataset_base.py:13-39) + per-sample file access (dataset_base.py:68-103)."""

    def __init__(self, root_dir: str, data_dir: str, split: str):
        if split not in ("train", "val", "test"):
            raise ValueError(f"Invalid split: {split}. Expected 'train', 'val', or 'test'.")
        with open(os.path.join(data_dir, f"{split}.json")) as f:
            ds = json.load(f)
        self.root_dir, self.split = root_dir, split
        self.id2data = {it["id"]: it for it in ds["images"]}
        self.annotations = ds["annotations"]

    def __len__(self):
        return len(self.annotations)

    def _take(self, data):
        if "take_name" in data:                               # data/README.md layout
            return data["take_name"]
        return os.path.join(data["dataset_name"], data["video_uid"])     # dataset_base.py:81-83

    def description(self, item: int) -> str:
        a = self.annotations[item]
        d = a.get("action_description", a.get("caption", ""))
        return d.lower() if isinstance(d, str) else d          # dataset_base.py:86-90

    def paths(self, item: int) -> dict:
        data = self.id2data[self.annotations[item]["image_id"]]
        take, fn = self._take(data), data["file_name"]
        r = self.root_dir
        tr = [os.path.join(r, "trajs", take, fn + e) for e in (".pkl", ".pickle", ".npz")]
        return {"image": os.path.join(r, "obs_images", take, fn + ".jpg"), "depth": os.path.join(r, "depths", take, fn + ".npy"),
                "pcrgb": os.path.join(r, "pcrgbs", take, fn + ".npy"), "traj": next((p for p in tr if os.path.exists(p)), tr[0])}

    def sample(self, item: int):
        """(image_id, pcrgb [N,6] float32, description, traj_rotvec [n,6])."""
        p = self.paths(item)
        pc = np.load(p["pcrgb"], allow_pickle=False).astype(np.float32)
        tr = load_traj_file(p["traj"])["traj_rotvec"]
        return self.annotations[item]["image_id"], pc, self.description(item), tr


class FileTrajData:
    """Adapter with the `.batch(idx, device, max_traj_token)` interface of driver.SyntheticTrajData,
    over files.  `encode(text) -> list[int]` is the caller's tokenizer (HF tokenizer in the reference);
    `norm` is the target normalisation (traj.TargetNorm: --do_norm / --do_standard, dataset.py:41-148)."""

    def __init__(self, dims, files: EgoScalerFiles, encode, num_steps=20, max_desc_token=20, smooth=False, norm=None, sep_ids=()):
        self.dims, self.files, self.encode, self.num_steps, self.max_desc, self.smooth = dims, files, encode, num_steps, max_desc_token, smooth
        self.norm = norm if norm is not None else T.TargetNorm(do_norm=True)
        self.sep_ids = tuple(sep_ids)                                   # ids of SEP_TOKEN between description and trajectory (dataset.py:54,169-175)

    def __len__(self):
        return len(self.files)

    def fit_norm(self, save_dir=None):
        """--do_standard on the train split: compute_mean_std over every trajectory file + norm_param.json (dataset.py:58-111)."""
        self.norm.fit([self.files.sample(i)[3] for i in range(len(self.files))], self.num_steps, self.smooth)
        if save_dir is not None:
            self.norm.save(save_dir)
        return self.norm

    def batch(self, idx, device, max_traj_token=160):
        return _file_batch(self, idx, device, max_traj_token)


def _pc_norm_np(pc: np.ndarray) -> np.ndarray:
    """pointllm/data/utils.py:146-157 on the host for clouds that arrive as files (dataset.py:14)."""
    xyz = pc[:, :3] - pc[:, :3].mean(0)
    return np.concatenate([xyz / np.sqrt((xyz ** 2).sum(1)).max(), pc[:, 3:]], 1)


def _file_batch(self, idx, device, max_traj_token=160):
    import torch
    from .driver import build_batch
    dims, N = self.dims, self.dims.pb.npoints
    pcs, descs, trs, gts, mabs, ids = [], [], [], [], [], []
    for i in idx:
        image_id, pc, desc, tr = self.files.sample(int(i))
        if pc.shape[0] < N:
            raise ValueError(f"sample {image_id}: cloud has {pc.shape[0]} points, need {N}")
        sel = np.arange(N) * (pc.shape[0] // N)                      # same deterministic stride as the clip glue
        pcs.append(_pc_norm_np(pc[sel].astype(np.float64)).astype(np.float32))
        descs.append(list(self.encode(desc))[: self.max_desc])       # --max_desc_token truncation (dataset.py:48)
        t = T.preprocess_traj(np.asarray(tr, dtype=np.float64), self.num_steps)           # traj_utils.py:3-39
        if self.smooth:
            t = T.smoothing_traj(t)
        v, m = self.norm.normalize(t)
        trs.append(np.clip(v, -1, 1).astype(np.float32))
        gts.append(t.astype(np.float32))
        mabs.append(m.astype(np.float32))
        ids.append(image_id)
    L = max(len(d) for d in descs)
    L += L % 2
    desc = np.full((len(descs), L), dims.tok.pad, dtype=np.int64)   # left-aligned; padding is masked out (dataset.py:161-177)
    dmask = np.zeros((len(descs), L), dtype=bool)
    for j, d in enumerate(descs):
        desc[j, :len(d)], dmask[j, :len(d)] = d, True
    dev = lambda a: torch.from_numpy(np.stack(a)).to(device)
    return build_batch(dims, torch.from_numpy(desc).to(device), dev(trs), dev(pcs), max_traj_token, image_ids=torch.as_tensor(ids, device=device),
                       desc_mask=torch.from_numpy(dmask).to(device), max_abs=dev(mabs), gt_trajs=dev(gts), sep_ids=self.sep_ids)
