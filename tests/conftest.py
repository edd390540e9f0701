import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """Compile the CPU oracle (test infrastructure) with g++."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    return os.path.join(ROOT, "oracle")
