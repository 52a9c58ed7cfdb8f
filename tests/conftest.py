import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A checkout without the built library (the .so is git-ignored): build it once, as __graft_entry__.build() does.
    The product itself never builds or falls back -- it raises if the library is missing."""
    so = os.path.join(ROOT, "ai-font-renderer_amd", "csrc", "libafr.so")
    if not os.path.exists(so):
        import shutil
        import subprocess
        if shutil.which("hipcc"):
            subprocess.run(["bash", os.path.join(ROOT, "ai-font-renderer_amd", "csrc", "build.sh")], check=False)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
