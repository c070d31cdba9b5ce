import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may sit silent for ever: a rank of a multi-process test that never answers, a wait for an event nobody records.
    With pytest-timeout installed every test without a limit of its own gets 420 s (the longest test here takes 60 s); when the
    limit strikes the plugin prints the stack of every thread, which is the evidence a hang otherwise never leaves."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(420))


@pytest.fixture(scope="session")
def gs():
    """The product package (its directory name is not a Python identifier)."""
    import importlib
    return importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
