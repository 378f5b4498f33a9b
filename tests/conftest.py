import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The HIP library must exist (built by __graft_entry__.build / python -m flowfusion_amd.build)."""
    from flowfusion_amd import _native
    if not _native.library_path().exists():
        from flowfusion_amd.build import build
        build()
    return _native.lib()
