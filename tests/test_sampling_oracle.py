"""CPU: oracle/sampling.py against the golden recorded from HF's own logits processors (tests/golden/sampling.npz, made by
oracle/gen_golden.py::gen_sampling from the reference model's logits), and the restated Philox4x32-10 against the published
known-answer vectors of Random123 (Salmon et al., SC'11, kat_vectors)."""
import os

import numpy as np
import torch

from oracle import sampling as OS


def cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampling.npz"), allow_pickle=False)
    for name in g["case_names"].tolist():
        T, k, p, rp = g[f"{name}:params"].tolist()
        yield name, torch.from_numpy(g[f"{name}:logits"]), torch.from_numpy(g[f"{name}:input_ids"]), torch.from_numpy(g[f"{name}:scores"]), T, int(k), p, rp


def test_processors_match_hf_golden(golden_dir):
    n = 0
    for name, lg, ids, want, T, k, p, rp in cases(golden_dir):
        got = OS.process(lg, ids, rp, T, k, p)
        ok, why = OS.same_up_to_boundary_ties(want, got, OS.process(lg, ids, rp, T, k, 1.0))
        assert ok, (name, why)
        n += 1
    assert n >= 15


def test_tie_equivalence_is_not_vacuous():
    w = torch.tensor([[1.0, float("-inf"), 2.0, 2.0, float("-inf")]])
    assert OS.same_up_to_boundary_ties(w, torch.tensor([[1.0, float("-inf"), 2.0, 2.0, float("-inf")]]))[0]
    pre = torch.tensor([[1.0, 0.5, 2.0, 2.0, 0.1]])
    assert not OS.same_up_to_boundary_ties(w, torch.tensor([[float("-inf"), 1.0, 2.0, 2.0, float("-inf")]]), pre)[0]      # index 1 never held 1.0
    assert not OS.same_up_to_boundary_ties(w, torch.tensor([[1.0, float("-inf"), 2.0, float("-inf"), float("-inf")]]))[0]      # fewer kept
    assert not OS.same_up_to_boundary_ties(w, torch.tensor([[1.5, float("-inf"), 2.0, 2.0, float("-inf")]]))[0]                # a value changed
    assert OS.same_up_to_boundary_ties(torch.tensor([[float("-inf"), 3.0, 3.0, 5.0]]), torch.tensor([[3.0, float("-inf"), 3.0, 5.0]]))[0]
    assert not OS.same_up_to_boundary_ties(torch.tensor([[float("-inf"), 3.0, 3.0, 5.0]]), torch.tensor([[3.0, 3.0, 3.0, float("-inf")]]))[0]


def test_philox4x32_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        r = OS.philox4x32(*[np.array([x]) for x in c], *k)
        assert tuple(int(x[0]) for x in r) == want


def test_gumbel_noise_is_standard_gumbel():
    g = OS.gumbel_noise(4, 20000, seed=1234, counter=7).astype(np.float64).ravel()
    assert abs(g.mean() - 0.5772) < 0.02 and abs(g.var() - np.pi ** 2 / 6) < 0.05
    assert not np.array_equal(OS.gumbel_noise(1, 64, 1234, 7), OS.gumbel_noise(1, 64, 1234, 8))
