"""Oracle for the token-choice half of generation: what HF GenerationMixin.generate does between two
forward passes under the arguments the reference passes (models/pointllm/model_arch.py:82-108: do_sample=True, top_k=50, top_p=0.95,
temperature=1.0, repetition_penalty, output_scores=True; train.py:223-228 validates this way).  Test infrastructure only (see
oracle/__init__.py).

The arithmetic lives in HuggingFace `transformers` (third-party, NOT under /root/reference; installed 5.15.0), restated here from
its published algorithm, transformers/generation/logits_process.py:
  :306-366  RepetitionPenaltyLogitsProcessor   score <- score * p if score < 0 else score / p, for every token already in input_ids
  :238-300  TemperatureLogitsWarper            scores / T
  :542-580  TopKLogitsWarper                   remove scores < (k-th largest value)          (ties with the k-th value are kept)
  :473-540  TopPLogitsWarper                   ascending sort, softmax, cumsum; remove the prefix with cumulative probability
                                               <= 1 - top_p; always keep the last `min_tokens_to_keep` = 1
applied in that order (generation/utils.py `_get_logits_processor`).  Pinned by tests/golden/sampling.npz, which oracle/gen_golden.py
records by calling HF's own processor classes (built by HF's own `_get_logits_processor`) on the reference model's logits.

The draw itself (torch.multinomial in HF) cannot be pinned — it consumes the global torch RNG.  The build draws with the Gumbel-max
trick on Philox4x32-10 bits (egoscaler_amd/csrc/sample.hip); `gumbel_noise` below restates that generator in numpy so that a test can
check the device's token against argmax(scores + noise) computed on the host.
"""
import numpy as np
import torch


def repetition_penalty(scores: torch.Tensor, input_ids: torch.Tensor, penalty: float) -> torch.Tensor:
    s = torch.gather(scores, 1, input_ids)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, input_ids, s)


def temperature(scores: torch.Tensor, T: float) -> torch.Tensor:
    return scores / T


def top_k(scores: torch.Tensor, k: int) -> torch.Tensor:
    k = min(max(int(k), 1), scores.shape[-1])
    kth = torch.topk(scores, k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, float("-inf"))


def top_p(scores: torch.Tensor, p: float) -> torch.Tensor:
    sv, si = torch.sort(scores, descending=False, stable=True)
    cum = sv.softmax(dim=-1).cumsum(dim=-1)
    rm = cum <= (1 - p)
    rm[..., -1:] = False
    return scores.masked_fill(rm.scatter(1, si, rm), float("-inf"))


def process(logits: torch.Tensor, input_ids=None, repetition_penalty_=1.0, temperature_=1.0, top_k_=50, top_p_=0.95) -> torch.Tensor:
    """fp32 [B,V] raw logits -> HF's processed scores (what `generate(..., output_scores=True).scores[t]` holds)."""
    s = logits.to(torch.float32)
    if repetition_penalty_ is not None and repetition_penalty_ != 1.0:
        s = repetition_penalty(s, input_ids, float(repetition_penalty_))
    if temperature_ is not None and temperature_ != 1.0:
        s = temperature(s, float(temperature_))
    if top_k_ is not None and top_k_ != 0:
        s = top_k(s, top_k_)
    if top_p_ is not None and top_p_ < 1.0:
        s = top_p(s, float(top_p_))
    return s


def same_up_to_boundary_ties(want: torch.Tensor, got: torch.Tensor, pre: torch.Tensor = None):
    """HF's top-p sorts with torch.sort(descending=False), which is NOT stable: among equal scores straddling the 1 - top_p boundary,
    which ones are removed depends on the sort implementation (CPU vs CUDA vs version; observed here: tokens 9 and 189 with equal score
    come out as 189, 9).  `top_p` above (and the HIP kernel) remove the lowest indices first.  Two results are therefore the same
    outcome iff, row by row, the kept VALUES are the same multiset (bit-exact) and the kept/removed pattern differs only inside ONE group
    of equal scores.  `pre` (optional): the scores in front of the top-p step (after penalty / temperature / top-k, none of which has
    any freedom): every kept value must then sit at its own position bit-exactly.  Returns (ok, message)."""
    if want.shape != got.shape:
        return False, "shape"
    for b in range(want.shape[0]):
        w, g = want[b], got[b]
        kw, kg = ~torch.isinf(w), ~torch.isinf(g)
        if not torch.equal(torch.sort(w[kw])[0], torch.sort(g[kg])[0]):
            return False, f"row {b}: kept values differ as multisets ({int(kw.sum())} vs {int(kg.sum())} kept)"
        diff = kw != kg
        if bool(diff.any()):
            vals = torch.where(kw, w, g)[diff]                        # the value each disputed token has where it was kept
            if float(vals.max()) != float(vals.min()):
                return False, f"row {b}: kept patterns differ outside one tie group: {vals.tolist()}"
        both = kw & kg
        if not torch.equal(w[both], g[both]):
            return False, f"row {b}: kept values differ"
        if pre is not None and not (torch.equal(g[kg], pre[b][kg]) and torch.equal(w[kw], pre[b][kw])):
            return False, f"row {b}: a kept value is not the score of its own position"
    return True, ""


# ---- the build's own draw (not a reference behaviour): Philox4x32-10 + Gumbel-max, restated from csrc/sample.hip
def philox4x32(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 numpy arrays -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n1, n2, n3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32, p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M32, (k1 + np.uint64(0xBB67AE85)) & M32
    return [x.astype(np.uint32) for x in (c0, c1, c2, c3)]


def gumbel_noise(B: int, V: int, seed: int, counter: int) -> np.ndarray:
    """fp32 [B,V]: the noise egomi_sample_rows adds to row b, column c when rng = (seed, counter base) and draw make `counter`."""
    c = np.arange(V, dtype=np.uint64)
    out = np.empty((B, V), dtype=np.float32)
    for b in range(B):
        r = philox4x32(c >> np.uint64(2), np.full(V, b), np.full(V, counter & 0xFFFFFFFF), np.full(V, (counter >> 32) & 0xFFFFFFFF),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        bits = np.stack(r, 0)[(c & np.uint64(3)).astype(np.int64), np.arange(V)]
        u = ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        out[b] = -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)
    return out
