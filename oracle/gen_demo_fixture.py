#!/usr/bin/env python3
"""tests/golden/demo_trajectory.npz + demo_info.json from the reference's demo sample (assets/demo/trajectory.pkl, info.json:
the one real EgoScaler record the release holds; written by data/train/7_get_object_trajectory.py:324-328, read by
models/utils/dataset_base.py:97-102).  Test infrastructure only; runs in the build container (needs /root/reference).

The pickle is NOT unpickled (serialized files that ship inside the reference are never loaded with anything that
unpickles): `pickletools.genops` only tokenises the opcode stream — no object is constructed, nothing named in the file
is imported or called.  The three float64 arrays are recovered from that token stream as data: dictionary keys are the
unicode tokens that are not numpy-internal names, the shape is the integer run that follows the state tuple's version
token, the payload is the bytes token.  info.json is JSON text and is copied as data.
"""
import json
import os
import pickletools

import numpy as np

REF = "/root/reference/assets/demo"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NUMPY_NAMES = {"numpy.core.multiarray", "numpy._core.multiarray", "_reconstruct", "numpy", "ndarray", "dtype", "f8", "<", "|", "="}


def arrays_from_pickle_tokens(path):
    ops = list(pickletools.genops(open(path, "rb").read()))
    out, key, dtypes = {}, None, set()
    for i, (op, arg, _) in enumerate(ops):
        if op.name in ("SHORT_BINUNICODE", "BINUNICODE"):
            if arg in NUMPY_NAMES:
                if arg not in ("numpy.core.multiarray", "numpy._core.multiarray", "_reconstruct", "numpy", "ndarray", "dtype"):
                    dtypes.add(arg)
            else:
                key = arg
        elif op.name in ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8") and len(arg) >= 16:
            # walk back to the state tuple of this array: MARK, BININT1 1 (version), <shape ints>, TUPLE*
            j = i
            while not (ops[j][0].name == "MARK" and ops[j + 1][0].name.startswith("BININT") and ops[j + 1][1] == 1 and ops[j + 2][0].name.startswith("BININT")):
                j -= 1
            shape, k = [], j + 2
            while ops[k][0].name.startswith("BININT"):
                shape.append(int(ops[k][1]))
                k += 1
            assert ops[k][0].name.startswith("TUPLE"), ops[k][0].name
            a = np.frombuffer(arg, dtype="<f8")
            assert a.size == int(np.prod(shape)), (key, shape, a.size)
            out[key] = a.reshape(shape).copy()
    assert dtypes <= {"f8", "<"}, dtypes                        # every array in the file is little-endian float64
    return out


if __name__ == "__main__":
    arrs = arrays_from_pickle_tokens(os.path.join(REF, "trajectory.pkl"))
    assert set(arrs) == {"init_bbox", "traj", "traj_rotvec"}, sorted(arrs)
    assert arrs["init_bbox"].shape == (8, 3) and arrs["traj"].shape[1] == 7 and arrs["traj_rotvec"].shape[1] == 6
    np.savez(os.path.join(OUT, "demo_trajectory.npz"), **arrs)
    info = json.load(open(os.path.join(REF, "info.json")))
    json.dump(info, open(os.path.join(OUT, "demo_info.json"), "w"), indent=1)
    print({k: v.shape for k, v in arrs.items()}, info["file_name"], info["take_name"])
