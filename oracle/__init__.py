"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package.
Nothing under egoscaler_amd/ imports it; the product path has no CPU fallback and raises when the
HIP library is missing.

Parity status: PINNED by outputs of the reference itself.  `oracle/gen_golden.py` imports the
reference's own modules from /root/reference (PointBERT, PointLLM model, model_arch) in the build
container, runs them on seeded synthetic weights/inputs, and commits the input/output vectors under
tests/golden/.  `tests/test_oracle_golden.py` checks every function here against those vectors.
Pieces of the reference that cannot be imported as released (utils/utils.py, dataset.py — ordinary
Python errors) are restated from their source text and pinned by known-answer
vectors computed from numpy/scipy semantics; they are marked "restated-from-text" in place.

The LLaMA arithmetic (RMSNorm/RoPE/attention/SwiGLU) lives in HuggingFace `transformers`, which is
third-party and not vendored under /root/reference; the version installed here (5.15.0) is what the
reference's classes subclass when imported, so the golden vectors include it.
"""
