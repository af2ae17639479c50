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
    spills), and one full-suite run in round 3 ended in a silent runtime abort inside a later, small test."""
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
