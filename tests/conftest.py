import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module("x2-vlm_amd")


@pytest.fixture(scope="session")
def synthetic():
    return importlib.import_module("x2-vlm_amd.synthetic")
