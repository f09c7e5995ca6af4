import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# the package directory name has hyphens: import it by string and alias it
rtvm = importlib.import_module("real-time-video-mosaic_amd")
sys.modules.setdefault("rtvm", rtvm)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _gpu_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ctx_small():
    """A context with no frame/canvas: matcher and RANSAC stages only."""
    c = rtvm.Context()
    yield c
    c.close()
