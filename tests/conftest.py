import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_cores() -> int:
    """cores really available (affinity, cgroup quota), capped at 16: os.cpu_count() reports the
    whole host on the GPU box and oversubscribes torch's CPU thread pool."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch

        torch.set_num_threads(host_cores())
    except Exception:
        pass


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["exact", "split"])
def f32_mode(request):
    """float32 tile GEMMs: exact f32 products on the f32 MFMA (default) or the opt-in three-term bf16 split
    (wipa_gemm_desc.f32_split / wipa_model_cfg.f32_split).  Tests pass ``f32_split=(f32_mode == "split")``."""
    return request.param
