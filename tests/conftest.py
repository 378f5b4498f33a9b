import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _usable_cores():
    """Affinity mask capped by the cgroup CPU quota: the GPU boxes expose 256 hardware threads but
    grant 16 cores; letting torch spawn 256 workers makes the CPU oracle several times slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope="session")
def built_library():
    """The HIP library must exist (built by __graft_entry__.build / python -m flowfusion_amd.build)."""
    from flowfusion_amd import _native
    if not _native.library_path().exists():
        from flowfusion_amd.build import build
        build()
    return _native.lib()
