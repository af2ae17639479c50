#!/usr/bin/env python3
"""BASELINE.json config 4: B=8 clips of 16 frames at 448x448 (3.2 M candidate pixels per sample) + the
point branch, rows A1-A6 timed separately with HIP events, achieved GB/s against the algorithmic
bytes of SURVEY.md §8d ("Roofline - gather/scan part").  GPU box only.  Also run by `python bench.py` after the headline
measurement (config.extra.pointbranch)."""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from egoscaler_amd import ops, synth
from egoscaler_amd.config import dims_7b

B, T, H, W = 8, 16, 448, 448


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3, out


def run(model=None, with_n4=True):
    """-> the JSON record (dict).  model: an existing bf16 TrajPointLLMForCausalLM whose (frozen, eval) point backbone is timed; None builds
    a small one around the full-size PointBERT."""
    dev = torch.device("cuda")
    if model is None:
        from egoscaler_amd.pointllm import TrajPointLLMForCausalLM
        dims = dims_7b()
        dims.lm.num_hidden_layers, dims.lm.hidden_size, dims.lm.intermediate_size, dims.lm.vocab_size, dims.lm.num_attention_heads = 1, 128, 128, 512, 1
        args = types.SimpleNamespace(unfreeze_pc_encoder=False, unfreeze_language_model=False, num_bins=16, model_name=None)
        model = TrajPointLLMForCausalLM(args, dims, None, device=dev, dtype=torch.bfloat16)
        sd = synth.synth_state_dict(dims, 0)
        model.load_state_dict({k: (v.bfloat16() if v.dtype.is_floating_point else v) for k, v in sd.items()})
    dims, eng = model.dims, model.engine
    g = np.random.Generator(np.random.Philox(key=np.array([4, 4], dtype=np.uint64)))
    rgb = torch.from_numpy(g.integers(1, 256, size=(B, T, H, W, 3), dtype=np.uint8))
    rgb[torch.from_numpy(g.random(size=(B, T, H, W)) < 0.05)] = 0
    depth = torch.from_numpy(g.uniform(0.3, 6.0, size=(B, T, H, W)).astype(np.float32))
    rgb, depth = rgb.to(dev), depth.to(dev)
    fx, pp = synth.clip_intrinsics(H)
    N, G, K = dims.pb.npoints, dims.pb.num_group, dims.pb.group_size
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    res = {}
    t, (pts, col, cnt) = timed(lambda: ops.unproject_gather(rgb, depth, pp, fx, fx, synth.DEPTH_THRESHOLD, n_out=N))
    px = B * T * H * W
    res["A1_unproject_subsample"] = {"ms": round(t * 1e3, 3), "alg_bytes": 7 * px + 36 * B * N, "GBps": round((7 * px + 36 * B * N) / t / 1e9, 1), "valid_px_min": int(cnt.min())}
    t, (ptsd, cold, cntd) = timed(lambda: ops.unproject_gather(rgb[:1], depth[:1], pp, fx, fx, synth.DEPTH_THRESHOLD), reps=5)
    nv = int(cntd[0])
    res["A1_unproject_dense_1sample"] = {"ms": round(t * 1e3, 3), "alg_bytes": 7 * T * H * W + 36 * nv, "GBps": round((7 * T * H * W + 36 * nv) / t / 1e9, 1), "n_valid": nv}
    if with_n4:
        # N4: depth map -> dense cloud at the Aria frame size (518^2 prediction -> 1408^2 frame), 8 frames
        Bn, h0n, Hn = 8, 518, 1408
        predn = torch.rand(Bn, h0n, h0n, device="cuda") * 4 + 0.2
        rgbn = torch.randint(0, 256, (Bn, Hn, Hn, 3), dtype=torch.uint8, device="cuda")
        t, _ = timed(lambda: ops.depth_to_cloud(predn, rgbn, Hn, Hn, 610.0, 610.0, 704), reps=5)
        bn = Bn * Hn * Hn * 59 + Bn * h0n * h0n * 4
        res["N4_depth_to_cloud"] = {"ms": round(t * 1e3, 3), "alg_bytes": bn, "GBps": round(bn / t / 1e9, 1), "note": "includes the three torch.empty outputs"}
        del predn, rgbn
    t, pc = timed(lambda: ops.pc_norm(pts, col))
    res["A2_pc_norm"] = {"ms": round(t * 1e3, 3), "alg_bytes": B * N * (36 + 24), "GBps": round(B * N * 60 / t / 1e9, 1)}
    t, (idx, cen) = timed(lambda: ops.fps(pc, start, G))
    res["A3_fps"] = {"ms": round(t * 1e3, 3), "alg_bytes": B * (12 * N + 16 * G), "on_chip_bytes": B * N * G * 16, "on_chip_TBps": round(B * N * G * 16 / t / 1e12, 2)}
    t, (kidx, nb) = timed(lambda: ops.knn_group(pc, cen, K, out_dtype=torch.bfloat16))
    res["A4_A5_knn_group"] = {"ms": round(t * 1e3, 3), "alg_bytes": B * (12 * N + 12 * G + 12 * G * K + 4 * G * K), "pairs_per_s_G": round(B * G * N / t / 1e9, 1)}
    t, tok = timed(lambda: eng.pointnet(nb.view(B * G * K, 6), B * G, K))
    pb = dims.pb
    fl = 2 * B * G * K * (6 * pb.pn_c1 + pb.pn_c1 * pb.pn_c2 + 2 * pb.pn_c2 * pb.pn_c3 + pb.pn_c3 * pb.encoder_dims)
    res["A6_pointnet"] = {"ms": round(t * 1e3, 3), "alg_flop": fl, "TFLOPs": round(fl / t / 1e12, 1)}
    t, feats = timed(lambda: eng.point_backbone(pc, start), reps=5)
    res["A3_A8_point_backbone_total"] = {"ms": round(t * 1e3, 3)}
    return {"config": "configs[3]: B=8, 16x448x448 RGB-D -> 8192-pt clouds + point branch, 1xMI355X, bf16", "rows": res}


if __name__ == "__main__":
    print(json.dumps(run()))
