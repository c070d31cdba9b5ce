import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gs():
    """The product package (its directory name is not a Python identifier)."""
    import importlib
    return importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
