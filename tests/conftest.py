import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="module", autouse=True)
def _release_cached_device_memory(request):
    """GPU box: hand the caching allocator's free blocks back to the driver after every test module.  The full-size modules reserve well over 100 GB
    each; what torch keeps cached is invisible to everything that allocates outside torch (RCCL, the runtime's scratch memory for kernels with
    spills), so modules start from a clean pool.  (This fixture was added while chasing round 3's intermittent abort of the full suite; cached
    memory was NOT its cause: `attn_bwd_dkdv2_kernel` read LSE / delta 128 B past the end of those arrays at S % 32 == 0 — commit 00c14e3,
    DESIGN.md §5 — and faulted whenever such an array closed a mapped segment.  tests/test_gpu_bounds.py now places every operand of the
    LDS-DMA kernels at the end of an allocation of its own, at aligned sizes too.)"""
    yield
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        log = os.environ.get("EGOMI_TEST_MEMLOG")
        if log:
            with open(log, "a") as f:
                f.write(f"{request.module.__name__} reserved_after_empty={torch.cuda.memory_reserved() >> 20} MiB "
                        f"peak_reserved={torch.cuda.max_memory_reserved() >> 20} MiB\n")
            torch.cuda.reset_peak_memory_stats()


@pytest.fixture(scope="session", autouse=True)
def _native_backtrace_on_abort():
    """EGOMI_ABORT_TRACE=1: SIGABRT prints the native stack of the aborting thread (tools/debug/abort_trace.c) before pytest's faulthandler dump."""
    if os.environ.get("EGOMI_ABORT_TRACE") == "1":
        import ctypes
        import subprocess
        import tempfile
        so = os.path.join(tempfile.mkdtemp(), "abort_trace.so")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tools", "debug", "abort_trace.c")])
        lib = ctypes.CDLL(so)
        lib.install_abort_trace()
        _ABORT_TRACE.append(lib)
    yield


_ABORT_TRACE = []


@pytest.fixture(autouse=True)
def _native_backtrace_on_abort_stays_installed():
    if _ABORT_TRACE:
        _ABORT_TRACE[0].install_abort_trace()          # (something re-registers SIGABRT during the suite)
    yield


@pytest.fixture(scope="module", autouse=True)
def _trace_device_allocations(request):
    """EGOMI_ALLOC_TRACE=<file>: inside test_gpu_train_modes / test_gpu_rccl_single every torch.empty / zeros (+ _like) of >= 256 KB on the device is logged
    as `start end bytes shape` — to match the address of a runtime 'Memory access fault' against the buffers that were alive (debug aid, round 3)."""
    path = os.environ.get("EGOMI_ALLOC_TRACE")
    name = request.module.__name__.rsplit(".", 1)[-1]
    if not path or name not in ("test_gpu_train_modes", "test_gpu_rccl_single"):
        yield
        return
    import torch
    f = open(path, "a")
    f.write(f"# module {name}\n")
    orig = {k: getattr(torch, k) for k in ("empty", "zeros", "empty_like", "zeros_like", "full")}

    def wrap(fn, k):
        def w(*a, **kw):
            t = fn(*a, **kw)
            if t.is_cuda and t.numel() * t.element_size() >= (256 << 10):
                n = t.numel() * t.element_size()
                f.write(f"{t.data_ptr():#x} {t.data_ptr() + n:#x} {n} {k} {tuple(t.shape)} {t.dtype}\n")
                f.flush()
            return t
        return w
    for k, fn in orig.items():
        setattr(torch, k, wrap(fn, k))
    try:
        yield
    finally:
        for k, fn in orig.items():
            setattr(torch, k, fn)
        f.close()
